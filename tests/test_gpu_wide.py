"""Worlds PAST one wavefront (round 6): more than 64 dofs or bodies run on the wide kernels (csrc/arb_wide_kernel.h: one
workgroup per world, float64 arithmetic) behind the same C ABI -- `arb_model_create` picks the path, `arb_model_info.wide`
says so.  The reference allocates any number of dofs (core.py:608-635), `add_snake(w, n)` takes any n
(robots/snake.py:17-60), human36 beside four free objects has 66.  Against the oracle, float64 (1e-8 where the oracle's own
explicit inverse allows it), through float32 and float64 buffers, one step and whole launches."""
import numpy as np
import pytest

import arb_oracle as O
from arboris_python_amd import _capi

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b), axis=-1) / np.maximum(1., np.max(np.abs(b), axis=-1))


@pytest.mark.parametrize("seed", list(range(1, 25)) + [1002, 1003, 1005, 1006, 1009, 1010])
def test_wide_random_trees_against_the_oracle(seed):
    """Random trees of 20-45 bodies and 65-200 dofs: every joint type, rotated frames on both sides of the joints, several
    roots, viscosity, spheres on a floor and on one another, a ball-and-socket loop closure, joint limits (the generator of
    tests/test_gpu_random_models.py with larger bounds): one step, float64, 1e-8 per world."""
    from test_gpu_random_models import random_world
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd import synth
    # (seeds past 1000: 56 - 62 constraints, 313 - 349 columns -- the compact build with six columns per lane)
    w = random_world(1000 + seed, nbody_range=(30, 44), max_dof=120, max_contacts=60, max_spheres=40) if seed > 1000 else \
        random_world(1000 + seed, nbody_range=(24, 46), max_dof=200, max_contacts=12, max_spheres=8)
    m, q0, dq0 = flatten_world(w)
    if m.ndof <= 64 and m.nb <= 64:
        pytest.skip("the generator came out small: %d dofs" % m.ndof)
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1 and bw.info["ndof"] == m.ndof
    B = 6
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1))
    qr, dqr = synth.random_states(m, B, seed=seed, angle=0.8, vel=1.5, root_box=((-.3, .3), (-.2, .5), (-.3, .3)))
    q[1:], dq[1:] = qr[1:], dqr[1:]
    dt = 2e-3
    cf0 = np.zeros((B, m.nc, 4))
    oq, odq, ocf = O.step(m, q, dq, dt, cforce=cf0)
    # (random states start with spheres deep in the floor and loop closures violated by tens of centimetres: |dq+| of
    # 1e2 .. 4e3 rad/s after one step; the metric is relative to it)
    ok = np.isfinite(oq).all(axis=1) & np.isfinite(odq).all(axis=1) & (np.abs(odq).max(axis=1) < 1e4)
    assert ok.sum() >= 1
    tq, tdq = bw.to_device(q, dq, torch.float64)
    tcf = bw.new_cforce(B, torch.float64)
    bw.step(tq, tdq, dt, 1, cforce=tcf)
    torch.cuda.synchronize()
    eq, edq = _rel(tq.cpu().numpy(), oq), _rel(tdq.cpu().numpy(), odq)
    print("seed %d: %d bodies, %d dofs, %d constraints (%s): err q %.2e dq %.2e (%d of %d worlds)"
          % (seed, m.nb, m.ndof, m.nc, sorted(set(m.ctype.tolist())), eq[ok].max(), edq[ok].max(), ok.sum(), B))
    assert eq[ok].max() < 1e-8 and edq[ok].max() < 1e-7
    ef = np.abs(tcf.cpu().numpy() - ocf)[ok].max() / max(1., np.abs(ocf[ok]).max())
    assert ef < 1e-6
    bw.close()


def test_snake_100_against_the_oracle_and_launch_shapes():
    """add_snake(w, 100): 100 Rz joints, a serial chain (cond(Z) ~ 1e9: the oracle's explicit inverse is itself good to ~1e-5,
    as for snake-64).  One step against the oracle; T steps in one launch == T one-step launches bit for bit; float32
    buffers == float64 buffers rounded (the arithmetic is float64 either way)."""
    from arboris_python_amd import scenes, synth
    from arboris_python_amd.batch import BatchedWorlds
    m = scenes.flat(scenes.snake_world(100))
    assert m.ndof == 100 and m.nb == 100
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1
    B, dt, T = 40, 1e-3, 5
    q, dq = synth.random_states(m, B, seed=0, angle=0.5, vel=1.0)
    oq, odq, _ = O.step(m, q[:8], dq[:8], dt)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    bw.step(tq, tdq, dt, 1)
    torch.cuda.synchronize()
    eq, edq = _rel(tq.cpu().numpy()[:8], oq), _rel(tdq.cpu().numpy()[:8], odq)
    print("snake-100 one step vs oracle: err q %.2e dq %.2e" % (eq.max(), edq.max()))
    assert eq.max() < 1e-7 and edq.max() < 5e-5
    # a better-conditioned check of the same kernel path at 1e-8: residual of the step equation with the oracle's matrices,
    #   (M/dt + N + B) dq+ = M dq/dt + gforce     (core.py:975-976), whose residual does not go through an inverse
    dyn = O.update_dynamic(m, q[:8], dq[:8])
    gforce, Z, Y = O.update_controllers(m, dyn, q[:8], dq[:8], dt)
    rhs = (dyn["M"] @ (dq[:8] / dt)[..., None])[..., 0] + gforce
    res = (Z @ tdq.cpu().numpy()[:8][..., None])[..., 0] - rhs
    res_o = (Z @ odq[..., None])[..., 0] - rhs
    print("residual |Z dq+ - rhs| / |rhs|: device %.2e, oracle %.2e" % (np.abs(res).max() / np.abs(rhs).max(), np.abs(res_o).max() / np.abs(rhs).max()))
    assert np.abs(res).max() / np.abs(rhs).max() < 1e-10
    a_q, a_dq = bw.to_device(q, dq, torch.float64)
    b_q, b_dq = bw.to_device(q, dq, torch.float64)
    bw.step(a_q, a_dq, dt, T)
    for _ in range(T):
        bw.step(b_q, b_dq, dt, 1)
    torch.cuda.synchronize()
    assert torch.equal(a_q, b_q) and torch.equal(a_dq, b_dq)
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    c_q, c_dq = bw.to_device(q, dq, torch.float32)
    d_q, d_dq = bw.to_device(f32(q), f32(dq), torch.float64)
    bw.step(c_q, c_dq, dt, T)
    bw.step(d_q, d_dq, dt, T)
    torch.cuda.synchronize()
    assert torch.equal(c_q, d_q.float()) and torch.equal(c_dq, d_dq.float())
    p = bw.plan(4096, 16, dtype=torch.float64)
    assert p["worlds_per_wavefront"] == 1 and p["wave_slots"] >= 256
    bw.close()


def test_human36_beside_four_free_objects_66_dofs():
    """human36 on its four floor contacts + four free boxes with a ball each on the floor: 66 dofs, 8 contacts.  A 30-step
    drop: every step replayed through the oracle from the device's own state (float64, 1e-8 / 1e-7), the contact forces of
    the last step; the trajectory log of the launch; an inspect of the first step (Z, contact activity)."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    w = scenes.human36_and_objects_world(4)
    m, q0, dq0 = flatten_world(w)
    assert m.ndof == 66 and m.nc == 8
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1
    B, dt, T = 16, 5e-3, 30
    rng = np.random.default_rng(3)
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + rng.uniform(-0.2, 0.2, (B, m.ndof))
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cf = bw.new_cforce(B, torch.float64)
    r = bw.inspect(tq, tdq, dt, ["Z", "c_active", "c_sdist", "gforce0", "q_next", "dq_next", "pose"], cforce=cf)
    dyn = O.update_dynamic(m, q, dq)
    gf, Zo, _ = O.update_controllers(m, dyn, q, dq, dt)
    assert np.abs(r["Z"].cpu().numpy() - Zo).max() / np.abs(Zo).max() < 1e-12
    assert _rel(r["gforce0"].cpu().numpy(), gf).max() < 1e-10
    assert np.abs(r["pose"].cpu().numpy() - dyn["pose"]).max() < 1e-12
    log = bw.rollout(tq, tdq, dt, T, cforce=cf, log_energy=False)
    torch.cuda.synchronize()
    lq, ldq = log["q"].cpu().numpy(), log["dq"].cpu().numpy()
    assert np.array_equal(lq[0], q) and np.array_equal(ldq[0], dq)
    assert _rel(r["dq_next"].cpu().numpy(), ldq[1]).max() < 1e-12
    worst = 0.
    ocf = None
    for t in range(T - 1):
        oq, odq, ocf = O.step(m, lq[t], ldq[t], dt)
        worst = max(worst, _rel(lq[t + 1], oq).max(), 0.1 * _rel(ldq[t + 1], odq).max())
    print("human36 + 4 objects: %d world-steps replayed, worst err (q, dq / 10) %.2e; max contact force %.0f N" % (B * (T - 1), worst, float(cf.abs().max())))
    assert worst < 1e-8
    assert float(cf.abs().max()) > 100.          # feet and balls are on the floor
    # user torques (a sequence) and a dense impedance on the same model: one launch against oracle steps
    tau = rng.uniform(-0.3, 0.3, (4, B, m.ndof)); tau[:, :, :6] = 0.
    za = -(0.2 * rng.uniform(-1, 1, (B, m.ndof, m.ndof)) + 1.5 * np.eye(m.ndof)[None])
    sq, sdq = bw.to_device(q, dq, torch.float64)
    scf = bw.new_cforce(B, torch.float64)
    bw.step(sq, sdq, dt, 4, cforce=scf, ext_gforce=torch.as_tensor(tau, device=bw.device).contiguous(),
            ext_impedance=torch.as_tensor(za, device=bw.device).contiguous())
    torch.cuda.synchronize()
    oq, odq, ocf = q, dq, None
    for t in range(4):
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf, ext_gforce=tau[t], ext_impedance=za)
    assert _rel(sq.cpu().numpy(), oq).max() < 1e-7 and _rel(sdq.cpu().numpy(), odq).max() < 1e-6
    # what the wide kernels do not take is refused, not ignored: the execution variants of the wavefront kernels, their diagnostics
    with pytest.raises(_capi.ArbError):
        bw.step(sq, sdq, dt, 1, cforce=scf, split="wave")
    with pytest.raises(_capi.ArbError):
        bw.inspect(sq, sdq, dt, ["gs_stats"])
    bw.close()


def test_limits_of_the_wide_path():
    """ndof, nb <= ARB_WIDE_MAX = 1024 (past 128 dofs the augmented system lives in scratch: a capability): snake-256 steps,
    snake-300 steps and satisfies the step equation, snake-1025 is refused with ARB_ERR_UNSUPPORTED (the reference would
    allocate it)."""
    import ctypes as C
    from arboris_python_amd import scenes, synth
    from arboris_python_amd.batch import BatchedWorlds
    assert _capi.ARB_WIDE_MAX == 1024
    for nl in (256, 300):
        m = scenes.flat(scenes.snake_world(nl))
        bw = BatchedWorlds(m)
        assert bw.info["wide"] == 1
        q, dq = synth.random_states(m, 3, seed=1, angle=0.3, vel=0.5)
        tq, tdq = bw.to_device(q, dq, torch.float64)
        dt = 1e-3
        bw.step(tq, tdq, dt, 1)
        torch.cuda.synchronize()
        assert torch.isfinite(tq).all() and torch.isfinite(tdq).all()
        # the step equation Z dq+ = M dq / dt + gforce (core.py:818-824) with the oracle's Z and right-hand side
        dyn = O.update_dynamic(m, q, dq)
        gf, Z, _ = O.update_controllers(m, dyn, q, dq, dt)
        rhs = (dyn["M"] @ dq[..., None])[..., 0] / dt + gf
        res = np.abs((Z @ tdq.cpu().numpy()[..., None])[..., 0] - rhs).max(axis=1) / np.abs(rhs).max(axis=1)
        print("snake-%d: residual of the step equation %.1e" % (nl, res.max()))
        assert res.max() < 1e-11
        bw.step(tq, tdq, dt, 2)
        torch.cuda.synchronize()
        assert torch.isfinite(tdq).all()
        bw.close()
    m2 = scenes.flat(scenes.snake_world(1025))
    desc, keep = _capi.make_desc(m2)
    h = C.c_void_p()
    assert _capi.load().arb_model_create(C.byref(desc), 0, C.byref(h)) == 2
    with pytest.raises(_capi.ArbError, match="1025 dofs"):       # (the Python layer says which limit)
        BatchedWorlds(m2)


@pytest.mark.parametrize("scene", ["objects4", "balls8"])
def test_object_api_simulates_a_wide_world(scene):
    """`simulate`'s loop body on a world of 66 dofs -- and on one of 90 dofs with the 108 contacts of every pair of
    `get_all_contacts` -- through the object API (World.update_dynamic / update_controllers / update_constraints / integrate on
    the device, a batch of one): body Jacobians, the world matrices M, B, N, the impedance and every step's state against the
    oracle (float64)."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world
    w = scenes.human36_and_objects_world(4) if scene == "objects4" else scenes.human36_and_balls_world(8)
    m, q, dq = flatten_world(w)
    dt = 5e-3
    w.update_dynamic()
    dyn = O.update_dynamic(m, q[None], dq[None])
    assert np.abs(w.mass - dyn["M"][0]).max() / np.abs(dyn["M"][0]).max() < 1e-12
    assert np.abs(w.nleffects - dyn["N"][0]).max() <= 1e-12 * max(1., np.abs(dyn["N"][0]).max())
    assert np.abs(w.viscosity - dyn["Bv"][0]).max() < 1e-12
    bodies = list(w.ground.iter_descendant_bodies())
    for b in (0, 5, 16, 17, 20):
        assert np.abs(bodies[b].jacobian - dyn["jac"][0, b]).max() < 1e-12 and np.abs(bodies[b].djacobian - dyn["djac"][0, b]).max() < 1e-11
        assert np.abs(bodies[b].twist - dyn["twist"][0, b]).max() < 1e-12
    oq, odq, ocf = q[None], dq[None], None
    for k in range(6):
        w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt); w.integrate(dt)
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf)
        qd = np.concatenate([np.asarray(j.gpos, float).ravel() for j in w.iterjoints()])
        assert _rel(qd, oq[0]) < 1e-9 and _rel(w.gvel, odq[0]) < 1e-8, k
    assert w._engine.flatten_count == 1


def test_wide_worlds_take_every_input_of_arb_step_ex():
    """Per-world PD targets and diagonal gains (a target SEQUENCE), the running cost with a torque sequence, energy logs: the wide
    kernels against the oracle (float64) on human36 + 4 objects."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world, JT_FREE
    from arboris_python_amd.batch import BatchedWorlds
    w = scenes.human36_and_objects_world(4)
    m, q0, dq0 = flatten_world(w)
    bw = BatchedWorlds(m)
    B, T, dt = 12, 5, 5e-3
    rng = np.random.default_rng(8)
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + rng.uniform(-0.2, 0.2, (B, m.ndof))
    free = np.zeros(m.ndof, bool)
    for b in range(m.nb):
        if m.jtype[b] == JT_FREE:
            free[int(m.dof_off[b]):int(m.dof_off[b]) + 6] = True
    kp = np.where(free, 0., rng.uniform(5., 30., (B, m.ndof))); kd = np.where(free, 0., rng.uniform(0.5, 2., (B, m.ndof)))
    qdes = rng.uniform(-0.2, 0.2, (T, B, m.ndof)); dqdes = rng.uniform(-0.1, 0.1, (T, B, m.ndof))
    tau = rng.uniform(-0.3, 0.3, (T, B, m.ndof)) * ~free
    host = dict(w_q=rng.uniform(0., 2., m.ndof), w_dq=rng.uniform(0., 0.1, m.ndof), w_tau=rng.uniform(0., 5., m.ndof), q_ref=rng.uniform(-0.2, 0.2, m.ndof))
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=bw.device)
    cost = {k: dev(v) for k, v in host.items()}
    cost["out"] = torch.zeros(B, dtype=torch.float64, device=bw.device)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cf = bw.new_cforce(B, torch.float64)
    log = bw.rollout(tq, tdq, dt, T, cforce=cf, ext_gforce=dev(tau), pd_targets=(dev(qdes), dev(dqdes)), pd_gains=(dev(kp), dev(kd)), cost=cost)
    torch.cuda.synchronize()
    oq, odq, ocf, ocost = q, dq, None, np.zeros(B)
    for t in range(T):
        dyn = O.update_dynamic(m, oq, odq)
        ke = 0.5 * np.einsum('bi,bij,bj->b', odq, dyn["M"], odq)
        pe = np.zeros(B)
        for b in range(m.nb):
            mb = m.mass[b][5, 5]
            if mb > 0:
                c = np.array([m.mass[b][2, 4], m.mass[b][0, 5], m.mass[b][1, 3]]) / mb
                cg = np.einsum('bij,j->bi', dyn["pose"][:, b, 0:3, 0:3], c) + dyn["pose"][:, b, 0:3, 3]
                pe += 9.81 * m.mass[b][3, 3] * (cg @ m.up)
        e = log["energy"][t].cpu().numpy()
        assert np.abs(e[:, 0] - ke).max() <= 1e-10 * max(1., np.abs(ke).max()) and np.abs(e[:, 1] - pe).max() <= 1e-10 * np.abs(pe).max(), t
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf, ext_gforce=tau[t], pd=dict(qdes=qdes[t], dqdes=dqdes[t], kp=kp, kd=kd))
        ocost += O.stage_cost(m, oq, odq, tau[t], **host)
    assert _rel(tq.cpu().numpy(), oq).max() < 1e-8 and _rel(tdq.cpu().numpy(), odq).max() < 1e-7
    assert np.abs(cost["out"].cpu().numpy() - ocost).max() <= 1e-10 * np.abs(ocost).max()
    bw.close()


@pytest.mark.parametrize("scene", ["snake100", "snake128", "snake140", "snake192", "human36+4", "human36+12", "human36+16", "random",
                                   "random:1003:100", "random:1005:100",
                                   "random:1010:100", "random:1003:200", "random:1016:200", "random:1018:200", "random:2002:many",
                                   "random:2005:many", "random:2010:many"])
def test_compact_build_equals_the_lds_build_bit_for_bit(scene):
    """Worlds of at most 192 dofs and 256 columns run the COMPACT build by default (arb_wide_kernel.h: the augmented system in
    registers, one LDS hand-over per pivot, log-depth chains shared with the other build); the knob "wide_compact" 0 selects the
    build that keeps the system in LDS / scratch.  Same assembly, same pivots, same multipliers, same sweeps: states, forces and
    every inspect output identical to the last bit, float32 and float64 buffers."""
    from arboris_python_amd import scenes, synth
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    if scene.startswith("snake"):                    # (snake-128: 129 columns -- four columns per lane instead of two; snake-140 / 192:
        m = scenes.flat(scenes.snake_world(int(scene[5:])))      #  40 / 48 rows per wavefront, the chain arrays of so many bodies in scratch)
        q, dq = synth.random_states(m, 12, seed=3, angle=0.5, vel=1.0)
        dt, steps = 1e-3, 6
    elif scene in ("human36+12", "human36+16"):      # (114 dofs, 16 contacts: 179 columns; 138 dofs, 20 contacts: 219 columns)
        m, q0, dq0 = flatten_world(scenes.human36_and_objects_world(int(scene[8:])))
        assert m.ndof == 42 + 6 * int(scene[8:]) and m.ndof + 1 + 4 * m.nc > 128
        q, dq = np.tile(q0, (12, 1)), np.tile(dq0, (12, 1))
        dq = dq + 0.05 * np.random.RandomState(2).standard_normal(dq.shape)
        dt, steps = 5e-3, 12
    elif scene == "human36+4":
        m, q0, dq0 = flatten_world(scenes.human36_and_objects_world(4))
        q, dq = np.tile(q0, (12, 1)), np.tile(dq0, (12, 1))
        dq = dq + 0.05 * np.random.RandomState(2).standard_normal(dq.shape)
        dt, steps = 5e-3, 25
    else:                                            # (random trees: contacts, loop closures, joint limits; 2 or 4 columns per lane)
        from test_gpu_random_models import random_world
        seed, big = (int(scene.split(":")[1]), scene.endswith(":200")) if ":" in scene else (1004, False)
        many = scene.endswith(":many")               # (56 - 62 constraints on 83 - 104 dofs: 313 - 349 columns, six per lane)
        w = random_world(seed, nbody_range=(30, 44), max_dof=120, max_contacts=60, max_spheres=40) if many else \
            random_world(seed, nbody_range=(24, 46), max_dof=200, max_contacts=12, max_spheres=8) if big else \
            random_world(seed, nbody_range=(24, 30), max_dof=100, max_contacts=6, max_spheres=4)
        m, q0, dq0 = flatten_world(w)
        ncols = m.ndof + 1 + 4 * m.nc
        assert 64 < m.ndof <= 128 and ((256 < ncols <= 384) if many else ((ncols > 128) == big)), (m.ndof, m.nc)
        q, dq = np.tile(q0, (12, 1)), np.tile(dq0, (12, 1))
        dt, steps = 2e-3, 3
    B = len(q)
    names = ["Z", "gforce0", "vel_free", "c_adm", "c_vel", "c_force", "gforce", "q_next", "dq_next", "pose", "twist"]
    for dtype in (torch.float64, torch.float32):
        out = {}
        for compact in (1, 0):
            bw = BatchedWorlds(m)
            assert bw.info["wide"] == 1
            bw.set_knob("wide_compact", compact)
            lds = bw.plan(B, steps, dtype=dtype)["lds_bytes"]
            tq, tdq = bw.to_device(q, dq, dtype)
            cf = bw.new_cforce(B, dtype) if m.nc else None
            r = bw.inspect(tq, tdq, dt, [n for n in names if m.nc or not n.startswith("c_")], cforce=cf)
            bw.step(tq, tdq, dt, steps, cforce=cf)
            torch.cuda.synchronize()
            out[compact] = (tq.cpu(), tdq.cpu(), None if cf is None else cf.cpu(), {k: v.cpu() for k, v in r.items()}, lds)
            bw.close()
        a, b = out[1], out[0]
        assert a[4] != b[4], "the two builds ask for different amounts of LDS: the knob did not switch"
        assert torch.isfinite(a[1]).all() or scene.startswith("random")     # (the generator's worlds are violent: a state may leave
        same = lambda x, y: torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x, nan=0.), torch.nan_to_num(y, nan=0.))
        assert same(a[0], b[0]) and same(a[1], b[1])                        #  the floats within three steps -- in both builds alike)
        if a[2] is not None:
            assert same(a[2], b[2])
        for k in a[3]:
            assert same(a[3][k], b[3][k]), k


@pytest.mark.parametrize("scene", ["human36+4", "human36+12", "random", "random:1003:100", "random:1010:100", "random:1003:200",
                                   "random:1014:200", "random:1016:200", "random:1028:200"])
def test_independent_groups_of_constraints_sweep_side_by_side_same_bits(scene):
    """The Gauss-Seidel sweeps of the wide kernels run the connected components of the constraint coupling (the non-zero 4 x 4
    blocks of the admittance) side by side -- the feet of human36 in one group, every object's contact in a group of its own:
    rounds of four solves instead of eight (sixteen with twelve objects).  The knob "wide_gs_groups" 0 runs the one serial
    sequence of core.py:929-935; states and forces are the same to the last bit over whole rollouts."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    if scene.startswith("human36+"):
        m, q0, dq0 = flatten_world(scenes.human36_and_objects_world(int(scene[8:])))
        steps, dt = 40, 5e-3
    else:
        # (random trees: spheres on the floor AND on one another -- groups that merge --, loop closures, joint limits)
        from test_gpu_random_models import random_world
        seed, big = (int(scene.split(":")[1]), scene.endswith(":200")) if ":" in scene else (1004, False)
        w = random_world(seed, nbody_range=(24, 46), max_dof=200, max_contacts=12, max_spheres=8) if big else \
            random_world(seed, nbody_range=(24, 30), max_dof=100, max_contacts=6, max_spheres=4)
        m, q0, dq0 = flatten_world(w)
        steps, dt = 3, 2e-3          # (the generator's worlds start with loop closures far open: violent, a few steps stay finite)
    B = 16
    q, dq = np.tile(q0, (B, 1)), np.tile(dq0, (B, 1))
    if scene != "random":
        dq = dq + 0.1 * np.random.RandomState(5).standard_normal(dq.shape)
    out = {}
    for groups in (1, 0):
        bw = BatchedWorlds(m)
        bw.set_knob("wide_gs_groups", groups)
        tq, tdq = bw.to_device(q, dq, torch.float64)
        cf = bw.new_cforce(B, torch.float64)
        bw.step(tq, tdq, dt, steps, cforce=cf)
        torch.cuda.synchronize()
        out[groups] = (tq.cpu(), tdq.cpu(), cf.cpu())
        bw.close()
    if not scene.startswith("random"):           # (the generator's worlds: violent, and contacts may all be open after three steps)
        assert torch.isfinite(out[1][1]).all() and out[1][2].abs().max() > 0
    for a, b in zip(out[1], out[0]):
        assert torch.equal(torch.nan_to_num(a, nan=1.25e300), torch.nan_to_num(b, nan=1.25e300))


def test_wide_kernels_against_the_reference_itself_g15():
    """tests/golden/g15_wide.npz holds what the REFERENCE computes on worlds past 64 dofs (tools/gen_golden.py: add_snake(w, 100)
    under gravity; human36 beside four free objects, 30 steps with contacts).  The wide kernels against it, float64: snake-100
    to the accuracy of the reference's own explicit inverse (cond(Z) ~ 1e9), the 66-dof scene step by step -- states, active
    sets through the constraint forces -- and as one 30-step launch."""
    from conftest import load_golden, load_model
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g15_wide.npz")
    m, _, _ = load_model("snake100_g")
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1
    dt = float(g["snake_dt"])
    tq, tdq = bw.to_device(g["snake_q"], g["snake_dq"], torch.float64)
    r = bw.inspect(tq, tdq, dt, ["Z"])
    Z0 = g["snake_Z0"]
    assert np.abs(r["Z"][0].cpu().numpy() - Z0).max() / np.abs(Z0).max() < 1e-12
    bw.step(tq, tdq, dt, 1)
    # cond(Z) ~ 1e10: the reference's dq+ = inv(Z) (M dq / dt + gforce) carries the error of numpy's explicit inverse (the oracle,
    # the same numpy calls, reproduces it to the bit).  The kernel's elimination is the more accurate of the two: both within
    # cond * eps of one another, and the kernel's dq+ satisfies the step equation with the smaller residual
    dyn = O.update_dynamic(m, g["snake_q"], g["snake_dq"])
    gf, Z, _ = O.update_controllers(m, dyn, g["snake_q"], g["snake_dq"], dt)
    rhs = (dyn["M"] @ g["snake_dq"][..., None])[..., 0] / dt + gf
    res = lambda x: (np.abs((Z @ x[..., None])[..., 0] - rhs).max(axis=1) / np.abs(rhs).max(axis=1))
    r_dev, r_ref = res(tdq.cpu().numpy()), res(g["snake_dq_next"])
    print("snake-100 step equation residual: kernel %.1e, reference %.1e; |kernel - reference| %.1e"
          % (r_dev.max(), r_ref.max(), _rel(tdq.cpu().numpy(), g["snake_dq_next"]).max()))
    assert r_dev.max() < 1e-12 and (r_dev < r_ref).all()
    assert _rel(tq.cpu().numpy(), g["snake_q_next"]).max() < 5e-7 and _rel(tdq.cpu().numpy(), g["snake_dq_next"]).max() < 5e-4
    tq, tdq = bw.to_device(g["snake_q"][:2], g["snake_dq"][:2], torch.float64)
    bw.step(tq, tdq, dt, 5)
    assert _rel(tq.cpu().numpy(), g["snake_roll5_q"]).max() < 3e-6 and _rel(tdq.cpu().numpy(), g["snake_roll5_dq"]).max() < 2e-3
    bw.close()
    for key, nobj in (("human", 4), ("human12", 12)):     # (66 dofs / 8 contacts: two columns per lane; 114 dofs / 16 contacts: four)
        m, _, _ = load_model("human36_obj%d" % nobj)
        bw = BatchedWorlds(m)
        assert bw.info["wide"] == 1 and bw.info["ndof"] == 42 + 6 * nobj
        dt = float(g[key + "_dt"])
        T = len(g[key + "_active"])
        tq, tdq = bw.to_device(g[key + "_q"][:1], g[key + "_dq"][:1], torch.float64)
        cf = bw.new_cforce(1, torch.float64)
        worst = 0.
        for k in range(T):
            bw.step(tq, tdq, dt, 1, cforce=cf)
            f = cf.cpu().numpy()[0]
            assert ((np.abs(f).max(axis=1) > 0) <= g[key + "_active"][k]).all(), k       # (a force only where the reference's contact is active)
            assert np.abs(f - g[key + "_force"][k]).max() <= 1e-7 * max(1., np.abs(g[key + "_force"][k]).max()), k
            e = max(_rel(tq.cpu().numpy(), g[key + "_q"][k + 1:k + 2]).max(), _rel(tdq.cpu().numpy(), g[key + "_dq"][k + 1:k + 2]).max())
            worst = max(worst, e)
        print("human36 + %d objects, %d steps against the reference: worst state error %.2e" % (nobj, T, worst))
        assert worst < 1e-8
        tq2, tdq2 = bw.to_device(g[key + "_q"][:1], g[key + "_dq"][:1], torch.float64)
        cf2 = bw.new_cforce(1, torch.float64)
        bw.step(tq2, tdq2, dt, T, cforce=cf2)
        assert torch.equal(tq2, tq) and torch.equal(tdq2, tdq) and torch.equal(cf2, cf)
        bw.close()


@pytest.mark.parametrize("nballs", [3, 8])
def test_every_pair_of_get_all_contacts_runs_wide(nballs):
    """`get_all_contacts` (constraints.py:840-875) pairs every two shapes of a world: human36 (eight foot points) beside free balls
    on a ground plane.  Three balls: 38 SoftFingerContacts -- plane / point, plane / ball, ball / point, ball / ball -- on 60 dofs,
    213 columns: past the wavefront kernels' 128 although the world has fewer than 64 dofs.  Eight balls: 108 contacts on 90 dofs
    -- more than the 64 a step may have ACTIVE, which is all that counts: the wide kernels solve on the step's active constraints
    (slots; `ARB_WIDE_MAX_CONSTRAINTS` = 256 may be registered).  25 steps against the oracle, step by step, float64; no warning."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    w = scenes.human36_and_balls_world(nballs)
    m, q0, dq0 = flatten_world(w)
    assert m.ndof == 42 + 6 * nballs and m.nc == 8 + nballs + nballs * (nballs - 1) // 2 + 8 * nballs
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1
    B, dt, T = 4, 5e-3, 25
    q = np.tile(q0, (B, 1))
    dq = np.tile(dq0, (B, 1)) + 0.1 * np.random.RandomState(7).standard_normal((B, m.ndof))
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cf = bw.new_cforce(B, torch.float64)
    oq, odq, ocf = q.copy(), dq.copy(), np.zeros((B, m.nc, 4))
    worst, seen = 0., np.zeros(m.nc, bool)
    for k in range(T):
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf)
        bw.step(tq, tdq, dt, 1, cforce=cf)
        worst = max(worst, _rel(tq.cpu().numpy(), oq).max(), _rel(tdq.cpu().numpy(), odq).max())
        seen |= (np.abs(ocf).max(axis=(0, 2)) > 0)
    print("human36 + %d balls, every pair (%d contacts): %d steps, worst state error %.2e, %d contacts carried a force"
          % (nballs, m.nc, T, worst, seen.sum()))
    assert worst < 1e-8 and seen.sum() >= 8
    assert np.abs(cf.cpu().numpy() - ocf).max() <= 1e-6 * max(1., np.abs(ocf).max())
    assert bw.warnings() == 0
    # the inspect outputs come back by CONSTRAINT, zero where one is not active
    r = bw.inspect(tq, tdq, dt, ["c_active", "c_adm", "c_force"], cforce=cf)
    act = r["c_active"][0].cpu().numpy().astype(bool)
    adm = r["c_adm"][0].cpu().numpy().reshape(m.nc, 4, m.nc, 4)
    assert 0 < act.sum() <= 64 and (np.abs(adm[~act]).max() == 0 if (~act).any() else True)
    assert (np.abs(adm[act][:, :, act]).max(axis=(1, 3)).diagonal() > 0).all()
    bw.close()


def test_more_than_64_active_constraints_raise_a_warning():
    """Seventy free balls resting on the floor: 70 plane / ball contacts active at once (420 dofs: the scratch build).  A step keeps
    the first 64 in registration order, leaves the others out of its solve -- their balls fall through -- and raises
    ARB_WARN_ACTIVE_CONSTRAINTS on the handle (include/arbstep.h); with 64 balls nothing is raised."""
    from arboris_python_amd.core import World, Body
    from arboris_python_amd.joints import FreeJoint
    from arboris_python_amd.shapes import Sphere
    from arboris_python_amd import massmatrix, homogeneousmatrix as Hg
    from arboris_python_amd.robots.simpleshapes import add_groundplane
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.constraints import get_all_contacts
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    for nballs, expect in ((64, 0), (70, _capi.ARB_WARN_ACTIVE_CONSTRAINTS)):
        w = World()
        add_groundplane(w)
        for k in range(nballs):
            body = Body(name="Ball%d" % k, mass=massmatrix.sphere(0.1, 1.0))
            j = FreeJoint(name="BallRoot%d" % k)
            j.gpos = Hg.transl(0.3 * (k % 10), 0.1, 0.3 * (k // 10))
            w.add_link(w.ground, j, body)
            w.register(Sphere(body, 0.1, name="Ball%d" % k))
        w.register(WeightController())
        for c in get_all_contacts(w, friction_coeff=0.6):
            if type(c._shapes[0]).__name__ == "Plane":
                w.register(c)
        w.init()
        m, q0, dq0 = flatten_world(w)
        assert m.nc == nballs and m.ndof == 6 * nballs
        bw = BatchedWorlds(m)
        tq, tdq = bw.to_device(q0[None], dq0[None], torch.float64)
        cf = bw.new_cforce(1, torch.float64)
        bw.step(tq, tdq, 5e-3, 2, cforce=cf)
        torch.cuda.synchronize()
        assert bw.warnings() == expect
        f = cf.cpu().numpy()[0]
        assert (np.abs(f[:64]).max(axis=1) > 0).all() and (np.abs(f[64:]) == 0).all()
        vy = tdq.cpu().numpy()[0].reshape(nballs, 6)[:, 4]          # (free joint twist: angular 0..2, linear 3..5; y is up)
        assert (np.abs(vy[:64]) < 1e-6).all() and (vy[64:] < -0.05).all()
        bw.close()


def test_scratch_build_with_contacts_past_the_compact_range():
    """human36 beside 26 free objects: 198 dofs, 30 contacts -- past the 192 dofs of the compact build: the system lives in the
    scratch block, the constraint slots, sweeps and products are the shared code.  Eight steps against the oracle, float64."""
    from arboris_python_amd import scenes
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    m, q0, dq0 = flatten_world(scenes.human36_and_objects_world(26))
    assert m.ndof == 198 and m.nc == 30
    bw = BatchedWorlds(m)
    assert bw.info["wide"] == 1 and bw.plan(2, 1, dtype=torch.float64)["lds_bytes"] < 64 * 1024      # (no system in LDS or registers)
    B, dt, T = 2, 5e-3, 8
    q = np.tile(q0, (B, 1))
    dq = np.tile(dq0, (B, 1)) + 0.05 * np.random.RandomState(3).standard_normal((B, m.ndof))
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cf = bw.new_cforce(B, torch.float64)
    oq, odq, ocf = q.copy(), dq.copy(), np.zeros((B, m.nc, 4))
    worst = 0.
    for k in range(T):
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf)
        bw.step(tq, tdq, dt, 1, cforce=cf)
        worst = max(worst, _rel(tq.cpu().numpy(), oq).max(), _rel(tdq.cpu().numpy(), odq).max())
    print("human36 + 26 objects (198 dofs, 30 contacts), scratch build: %d steps, worst state error %.2e" % (T, worst))
    assert worst < 1e-8 and np.abs(ocf).max() > 0
    assert np.abs(cf.cpu().numpy() - ocf).max() <= 1e-6 * max(1., np.abs(ocf).max())
    bw.close()
