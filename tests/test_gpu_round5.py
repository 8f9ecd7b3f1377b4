"""Round-5 GPU tests (through the C ABI, ABI 7).

* control inputs that change along the horizon: the reference polls every controller EVERY step (core.py:811-817,
  controllers.py:141-158), so an MPC rollout applies a torque SEQUENCE -- `arb_step_args.ext_gforce_steps`,
  `pd_qdes_steps` / `pd_dqdes_steps`, read step by step inside ONE launch -- and returns a cost per rollout
  (`arb_step_cost`), summed on chip;
* `arb_model_warnings` / ARB_WARN_ILLCOND: float32 on a model whose impedance matrix float32 cannot eliminate
  (snake-64, BASELINE config 4's model: velocity error 0.25 with ARB_OK until round 4) now says so;
* the library reads no environment variable: the development knobs go through `arb_hook_set_knob`.
"""
import numpy as np
import pytest

from conftest import load_model, load_golden
from arboris_python_amd import _capi
import arb_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

F32_TOL = 1e-5          # the north star's float32 tolerance (relative, per world: max|x - ref| / max(1, max|ref|))


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b), axis=-1) / np.maximum(1., np.max(np.abs(b), axis=-1))


def _torque_sequence(m, T, B, seed, amp=0.05, base=6):
    """Smooth, time-varying joint torques: a different sinusoid per rollout and dof, none on the floating base."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-amp, amp, size=(1, B, m.ndof))
    ph = rng.uniform(0., 2. * np.pi, size=(1, B, m.ndof))
    om = rng.uniform(0.2, 1.0, size=(1, B, m.ndof))
    tau = a * np.sin(om * np.arange(T)[:, None, None] + ph)
    tau[:, :, :base] = 0.
    return tau


def _cost_tensors(bw, m, dtype, B, seed=3):
    rng = np.random.default_rng(seed)
    host = dict(w_q=rng.uniform(0., 2., m.ndof), w_dq=rng.uniform(0., 0.1, m.ndof), w_tau=rng.uniform(0., 5., m.ndof),
                q_ref=rng.uniform(-0.2, 0.2, m.ndof))
    dev = {k: torch.as_tensor(v, dtype=dtype, device=bw.device).contiguous() for k, v in host.items()}
    dev["out"] = torch.zeros(B, dtype=dtype, device=bw.device)
    return host, dev


@pytest.mark.parametrize("name,dtype,B,T,kw", [
    ("human36_c4", "float32", 64, 32, {}),                               # the MPC shape (the specialised kernels, FEAT 5)
    ("human36_c4", "float32", 64, 32, dict(general_kernels=True)),       # the general kernels (FEAT 1)
    ("human36_c4", "float32", 6000, 12, {}),                             # more worlds than wave slots: through the work queue
    ("human36_c4", "float64", 300, 16, {}),
    ("human36_c8", "float32", 200, 16, {}),                              # two column sets
    ("simplearm", "float32", 5000, 16, {}),                              # a forest of 8 copies per wavefront (no cost there)
    ("simplearm", "float32", 5003, 16, {}),                              # ... and three worlds left over, one per wavefront
])
def test_torque_sequence_in_one_launch_equals_one_step_launches_bitwise(name, dtype, B, T, kw):
    """`O.step(..., ext_gforce=tau_t)` for t = 0 .. T-1 inside ONE launch (`ext_gforce` of rank 3) == T one-step launches fed
    tau_t: states, forces AND the running cost bit for bit (the cost adds one step at a time in step order, however the
    horizon is cut into launches or work items)."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, q0, dq0 = load_model(name)
    dt_ = getattr(torch, dtype)
    bw = BatchedWorlds(m)
    if name.startswith("human36"):
        q, dq = synth.standing_states(m, B, seed=77, drop=0.03, vel=0.1)
        q[:, 7] -= 0.01
    else:
        rng = np.random.default_rng(1)
        q, dq = rng.uniform(-1., 1., (B, m.nq)), rng.uniform(-1., 1., (B, m.ndof))
    tau_h = _torque_sequence(m, T, B, seed=5) if name != "simplearm" else _torque_sequence(m, T, B, seed=5, amp=0.5, base=0)
    tau = torch.as_tensor(tau_h, dtype=dt_, device=bw.device).contiguous()
    with_cost = name != "simplearm"         # (a running cost is per world: with one the small model would not run as a forest)
    res = {}
    for mode in ("one_launch", "per_step", "two_launches"):
        tq, tdq = bw.to_device(q, dq, dt_)
        cf = bw.new_cforce(B, dt_) if m.nc else None
        _, cost = _cost_tensors(bw, m, dt_, B)
        ck = dict(cost=cost) if with_cost else {}
        if mode == "one_launch":
            bw.step(tq, tdq, 5e-3, T, cforce=cf, ext_gforce=tau, **ck, **kw)
        elif mode == "per_step":
            for t in range(T):
                bw.step(tq, tdq, 5e-3, 1, cforce=cf, ext_gforce=tau[t].contiguous(), **ck, **kw)
        else:
            h = T // 2 + 1
            bw.step(tq, tdq, 5e-3, h, cforce=cf, ext_gforce=tau[:h].contiguous(), **ck, **kw)
            bw.step(tq, tdq, 5e-3, T - h, cforce=cf, ext_gforce=tau[h:].contiguous(), **ck, **kw)
        torch.cuda.synchronize()
        bw.status()
        res[mode] = [tq, tdq, cost["out"]] + ([cf] if cf is not None else [])
    bits = torch.int32 if dtype == "float32" else torch.int64
    for mode in ("per_step", "two_launches"):
        assert all(torch.equal(a.view(bits), b.view(bits)) for a, b in zip(res["one_launch"], res[mode])), (mode, name, B)
    assert bool(torch.isfinite(res["one_launch"][0]).all())
    assert not with_cost or float(res["one_launch"][2].min()) > 0.        # (a cost was accumulated)
    if name == "simplearm":
        assert bw.plan(B, T, ext_gforce=True)["worlds_per_wavefront"] == bw.info["forest_copies"] > 1
    # the sequence matters: the same launch with the FIRST row for every step ends elsewhere
    tq, tdq = bw.to_device(q, dq, dt_)
    bw.step(tq, tdq, 5e-3, T, cforce=bw.new_cforce(B, dt_) if m.nc else None, ext_gforce=tau[0].contiguous(), **kw)
    torch.cuda.synchronize()
    assert float((tdq - res["one_launch"][1]).abs().max()) > 1e-4
    bw.close()


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-7), ("float32", F32_TOL)])
def test_mpc_rollouts_with_torque_sequences_and_costs_against_the_oracle(dtype, tol):
    """64 rollouts x a 32-step horizon with time-varying torques in ONE launch, replayed step by step through the oracle
    from the device's own logged states, `O.step(..., ext_gforce=tau_t)`: the next logged state within the tolerance
    (float32: >= 99.5 % of the world-steps, the others contact decisions as in config 3), and the per-rollout cost equals
    the oracle's quadratic form summed over the logged trajectory."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    dt_ = getattr(torch, dtype)
    bw = BatchedWorlds(m)
    B, T, dt = 64, 32, 5e-3
    q, dq = synth.standing_states(m, B, seed=9, drop=0.03, vel=0.1)
    tau_h = _torque_sequence(m, T, B, seed=11)
    tau = torch.as_tensor(tau_h, dtype=dt_, device=bw.device).contiguous()
    tq, tdq = bw.to_device(q, dq, dt_)
    host, cost = _cost_tensors(bw, m, dt_, B)
    log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, dt_), ext_gforce=tau, log_energy=False, cost=cost)
    torch.cuda.synchronize()
    lq = torch.cat([log["q"], tq[None]]).double().cpu().numpy()           # states 0 .. T (the last one: the final state)
    ldq = torch.cat([log["dq"], tdq[None]]).double().cpu().numpy()
    tau_d = tau.double().cpu().numpy()
    eq, edq = [], []
    ref_cost = np.zeros(B)
    for t in range(T):
        oq, odq, _ = O.step(m, lq[t], ldq[t], dt, ext_gforce=tau_d[t])
        eq.append(_rel(lq[t + 1], oq)); edq.append(_rel(ldq[t + 1], odq))
        # the cost the device accumulated: the state AFTER step t (its own), the torques of step t
        ref_cost += O.stage_cost(m, lq[t + 1], ldq[t + 1], tau_d[t], **{k: np.asarray(v, np.float32 if dtype == "float32" else np.float64).astype(np.float64) for k, v in host.items()})
    eq, edq = np.concatenate(eq), np.concatenate(edq)
    ok = (eq < tol) & (edq < tol)
    print("torque sequences vs oracle (%s): %d world-steps, ok %.4f, max err q %.2e dq %.2e" % (dtype, len(ok), ok.mean(), eq.max(), edq.max()))
    assert ok.mean() >= (0.995 if dtype == "float32" else 1.0) and eq.max() < 1e-2 and edq.max() < 1e-1
    got = cost["out"].double().cpu().numpy()
    assert np.max(np.abs(got - ref_cost) / np.maximum(1., np.abs(ref_cost))) < (2e-6 if dtype == "float32" else 1e-12)
    # the torques of step t are in the oracle's step t: with the torques of step 0 throughout it is off
    o1, od1, _ = O.step(m, lq[20], ldq[20], dt, ext_gforce=tau_d[0])
    assert _rel(ldq[21], od1).max() > 10 * F32_TOL
    bw.close()


def test_pd_target_sequence_equals_one_step_launches_and_the_oracle():
    """Per-step PD targets (`pd_qdes_steps` / `pd_dqdes_steps`): a posture servo that follows a moving target inside one
    launch == one-step launches fed the targets of the step, bit for bit; and against the oracle's PD controller
    (controllers.py:141-158) on the first steps."""
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g8_pd_per_world.npz")
    m, _, _ = load_model("simplearm_pdw")
    bw = BatchedWorlds(m)
    Q, DQ = g["arm_g_q"], g["arm_g_dq"]
    W, T = Q.shape[1], 24
    rng = np.random.default_rng(4)
    qdes = g["arm_g_qdes"][None] + 0.3 * np.sin(0.3 * np.arange(T)[:, None, None] + rng.uniform(0, 6, (1, W, m.ndof)))
    dqdes = np.zeros_like(qdes) + g["arm_g_dqdes"][None]
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, F32_TOL)):
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=bw.device).contiguous()
        kp, kd = dev(g["arm_g_kp"]), dev(g["arm_g_kd"])
        tqd, tdqd = dev(qdes), dev(dqdes)
        tq, tdq = bw.to_device(Q[0], DQ[0], dtype)
        bw.step(tq, tdq, 5e-3, T, pd_targets=(tqd, tdqd), pd_gains=(kp, kd), one_world=True)
        sq, sdq = bw.to_device(Q[0], DQ[0], dtype)
        oq, odq = np.asarray(sq.double().cpu()), np.asarray(sdq.double().cpu())
        worst = 0.
        for t in range(T):
            if t < 6:          # the oracle from the device's own state
                f = lambda a: np.asarray(a, np.float32 if dtype == torch.float32 else np.float64).astype(np.float64)
                rq, rdq, _ = O.step(m, sq.double().cpu().numpy(), sdq.double().cpu().numpy(), 5e-3,
                                    pd=dict(qdes=f(qdes[t]), dqdes=f(dqdes[t]), kp=f(g["arm_g_kp"]), kd=f(g["arm_g_kd"])))
            bw.step(sq, sdq, 5e-3, 1, pd_targets=(tqd[t].contiguous(), tdqd[t].contiguous()), pd_gains=(kp, kd), one_world=True)
            if t < 6:
                worst = max(worst, _rel(sq.double().cpu().numpy(), rq).max(), _rel(sdq.double().cpu().numpy(), rdq).max())
        torch.cuda.synchronize()
        assert torch.equal(tq, sq) and torch.equal(tdq, sdq)
        assert worst < tol, worst
        # a constant target gives another trajectory
        cq, cdq = bw.to_device(Q[0], DQ[0], dtype)
        bw.step(cq, cdq, 5e-3, T, pd_targets=(tqd[0].contiguous(), tdqd[0].contiguous()), pd_gains=(kp, kd), one_world=True)
        torch.cuda.synchronize()
        assert float((cq - tq).abs().max()) > 1e-3
    bw.close()


def test_step_ex_sequence_argument_validation():
    from arboris_python_amd.batch import BatchedWorlds
    m, q0, dq0 = load_model("human36_c4")
    bw = BatchedWorlds(m)
    B, T = 4, 3
    tq, tdq = bw.to_device(np.tile(q0, (B, 1)), np.tile(dq0, (B, 1)), torch.float32)
    z3 = torch.zeros((T, B, m.ndof), dtype=torch.float32, device=bw.device)
    with pytest.raises(ValueError):                                   # a sequence needs one row per step
        bw.step(tq, tdq, 5e-3, T + 1, ext_gforce=z3)
    with pytest.raises(ValueError):                                   # the cost needs its accumulator
        bw.step(tq, tdq, 5e-3, T, cost=dict(w_q=torch.zeros(m.ndof, device=bw.device)))
    with pytest.raises(_capi.ArbError):                               # no running cost with the split execution
        bw.step(tq, tdq, 5e-3, T, cforce=bw.new_cforce(B, torch.float32), split="wave",
                cost=dict(out=torch.zeros(B, device=bw.device)))
    # both a constant torque and a sequence: refused by the library
    a = _capi.StepArgs()
    a.q, a.dq, a.nworlds, a.nsteps, a.dt = tq.data_ptr(), tdq.data_ptr(), B, T, 5e-3
    a.ext_gforce, a.ext_gforce_steps = z3.data_ptr(), z3.data_ptr()
    assert bw._lib.arb_step_ex(bw._handle, _capi.ARB_F32, a, None) == 1
    assert bw._lib.arb_hook_set_knob(bw._handle, b"no_such_knob", 1) == 1
    torch.cuda.synchronize()
    bw.close()


def test_float32_on_an_ill_conditioned_model_is_right_by_default_and_warns_when_pinned():
    """snake-64 (BASELINE config 4's model) with float32 buffers.  Plain float32 elimination of its impedance matrix cancels
    ~17 of float32's 24 bits (pivot growth ~1e5): the velocities are wrong by tens of per cent, and `arb_model_warnings`
    says so (ARB_WARN_ILLCOND) -- that is `mixed=False` (ARB_STEP_NO_MIXED) since round 6.  BY DEFAULT the library now
    PROMOTES such a launch to the float64 kernels (arb_model_info.mixed_default == 2, from the pivot growth at the rest
    states, probed by arb_model_create; conversion kernels around the float64 launch): right to 1e-5, no warning.  The mixed
    build on request (`mixed=True`: float32 state, float64 elimination and right-hand side) is right to 2e-4 -- the body
    wrenches stay float32 and this model amplifies every rounding by up to 3e8.  In float64 the same launch is
    exact to 1e-9.  human36 (growth < 2^11) stays on the float32 kernels and never warns, on the ground or in free motion, one step
    or whole episodes through the work queue (as long as the world itself has not diverged)."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g4_snake64.npz")
    m, _, _ = load_model("snake64_g")
    bw = BatchedWorlds(m)
    assert bw.info["mixed_default"] == 2 and bw.info["rest_pivot_growth"] > _capi.ARB_ILLCOND_GROWTH
    q, dq = g["q"], g["dq"]
    gro = {}
    for dtype, mixed in ((torch.float32, False), (torch.float32, True), (torch.float32, None), (torch.float64, None)):
        tq, tdq = bw.to_device(q, dq, dtype)
        gro[dtype] = bw.inspect(tq, tdq, 1e-3, ["pivot_growth"])["pivot_growth"].double().cpu().numpy()
        assert bw.warnings() == 0                                          # (inspecting raises nothing)
        bw.step(tq, tdq, 1e-3, 1, mixed=mixed)
        torch.cuda.synchronize()
        w = bw.warnings()
        err = _rel(tdq.double().cpu().numpy(), g["dq_next"]).max()
        print("snake-64 %s mixed=%s: pivot growth %.3g .. %.3g, dq+ error %.2e, warnings %d" % (dtype, mixed, gro[dtype].min(), gro[dtype].max(), err, w))
        if dtype == torch.float32 and mixed is False:
            assert w == _capi.ARB_WARN_ILLCOND and gro[dtype].min() > _capi.ARB_ILLCOND_GROWTH and err > 1e-2
            assert bw.warnings() == 0                                      # reading cleared it
        elif dtype == torch.float32 and mixed is True:
            assert w == 0 and err < 2e-4
        else:
            assert w == 0 and err < 1e-5        # (the golden dq+ is the reference's explicit inverse: ~3e-6, tests/test_gpu_parity.py)
    # the float32 measure tracks the float64 one -- or SATURATES: in two of these eight worlds a float32 pivot comes out <= 0 (the
    # elimination has gone indefinite), which round 5's bit-pattern subtraction wrapped into "no growth" (arb_growth_bits)
    fin = np.isfinite(gro[torch.float32])
    assert np.isfinite(gro[torch.float64]).all() and fin.sum() >= 4
    assert np.all(gro[torch.float32][fin] > gro[torch.float64][fin] / 8.) and np.all(gro[torch.float32][fin] < gro[torch.float64][fin] * 8.)
    bw.close()
    for name in ("human36_c4", "human36_g", "human36_c8"):
        m, _, _ = load_model(name)
        bw = BatchedWorlds(m)
        assert bw.info["mixed_default"] == 0 and bw.info["rest_pivot_growth"] < _capi.ARB_ILLCOND_GROWTH / 8.
        B = 5000
        if m.nc:
            qh, dqh = synth.standing_states(m, B, seed=3, drop=0.03, vel=0.1)
        else:
            qh, dqh = synth.world_states(m, range(B), "random", 3, angle=0.7, vel=1.0)
        tq, tdq = bw.to_device(qh, dqh, torch.float32)
        gr = bw.inspect(tq[:512].contiguous(), tdq[:512].contiguous(), 5e-3, ["pivot_growth"])["pivot_growth"]
        print("%s: pivot growth up to %.1f" % (name, float(gr.max())))
        assert float(gr.max()) < _capi.ARB_ILLCOND_GROWTH / 4.
        cf = bw.new_cforce(B, torch.float32) if m.nc else None
        # (free motion from random states: 20 steps -- the reference's own time stepping sends a few of 5000 such worlds
        # beyond 1e5 rad/s by step 35, tools/growth_probe.py, and a diverging world IS ill-conditioned)
        bw.step(tq, tdq, 5e-3, 40 if m.nc else 20, cforce=cf)
        bw.step(tq, tdq, 5e-3, 1, cforce=cf)
        torch.cuda.synchronize()
        assert bw.warnings() == 0, name
        bw.close()


# ---------------------------------------------------------------------------
# body-space constraint columns (FEAT bit 16): the reference's own eight-contact scenario in ONE column set
# ---------------------------------------------------------------------------
def _contact_subset(m, keep):
    """The flattened model with the contacts `keep` only (a copy)."""
    import copy
    m2 = copy.deepcopy(m)
    keep = np.asarray(keep)
    for k, v in vars(m).items():
        if k == "c_names":
            m2.c_names = [m.c_names[i] for i in keep]
        elif (k.startswith("c_") or k == "ctype") and isinstance(v, np.ndarray) and len(v) == m.nc:
            setattr(m2, k, v[keep].copy())
    return m2


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_body_space_columns_against_the_oracle_and_the_two_set_kernels(dtype):
    """human36 with the reference's eight contact points (tests/test_human36_falling.py:32): all contacts of a foot are
    T_c J_foot (constraints.py:429-433), so the augmented system carries 6 columns per foot (12) instead of 4 per contact (32)
    and fits one column set; Y' = T (J Y J^T) T^T is formed after phase D.  One step from the reference's own drop trajectory
    and from random contact states, against the oracle (float64: 1e-8; float32: the 1e-5 gate with the usual adjudication of
    ill-conditioned steps) and against the general kernels on two column sets (ARB_STEP_GENERAL_KERNELS); the inspect
    kernel's Y', v', J' against the oracle's; a six-contact model (three per foot: 67 columns the classical way) likewise."""
    from conftest import assert_f32_parity
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g3_contacts.npz")
    m8, _, _ = load_model("human36_c8")
    dt_ = getattr(torch, dtype)
    npt = np.float64 if dtype == "float64" else np.float32
    Q = np.concatenate([g["drop8_q"][:39], g["rand8_q"]]); DQ = np.concatenate([g["drop8_dq"][:39], g["rand8_dq"]])
    for m, tag in ((m8, "eight"), (_contact_subset(m8, [0, 1, 2, 4, 5, 7]), "six")):
        bw = BatchedWorlds(m)
        p = bw.plan(8192, 40, dtype=dt_)
        assert p["feat"] == 20 and p["waves_per_simd"] == (3 if dtype == "float32" else 2), p
        assert bw.plan(8192, 40, dtype=dt_, general_kernels=True)["feat"] == 0
        if dtype == "float32" and tag == "eight":
            assert p["lds_bytes"] <= 10 * 1280 and p["wave_slots"] == 12 * torch.cuda.get_device_properties(0).multi_processor_count
        qi, dqi = Q.astype(npt).astype(np.float64), DQ.astype(npt).astype(np.float64)
        oq, odq, ocf, dbg = O.step(m, qi, dqi, 5e-3, debug=True)
        res = {}
        for gk in (False, True):
            tq, tdq = bw.to_device(Q, DQ, dt_)
            cf = bw.new_cforce(len(Q), dt_)
            bw.step(tq, tdq, 5e-3, 1, cforce=cf, general_kernels=gk)
            torch.cuda.synchronize()
            res[gk] = (tq.double().cpu().numpy(), tdq.double().cpu().numpy(), cf.double().cpu().numpy())
            if dtype == "float64":
                assert _rel(res[gk][0], oq).max() < 1e-8 and _rel(res[gk][1], odq).max() < 1e-8, (tag, gk)
                assert np.abs(res[gk][2] - ocf).max() < 1e-7 * max(1., np.abs(ocf).max())
            else:
                assert_f32_parity(m, Q, DQ, 5e-3, res[gk][0], res[gk][1], oq, odq, 1e-5)
        d = np.maximum(_rel(res[False][0], res[True][0]), _rel(res[False][1], res[True][1]))
        print("%s contacts, %s: body-space vs two column sets: max %.2e, share > 1e-5: %.4f" % (tag, dtype, d.max(), (d > 1e-5).mean()))
        # (float32: two roundings of the same step, each within ~5e-6 of the reference: a few pairs differ by just over 1e-5)
        assert d.max() < (1e-9 if dtype == "float64" else 1e-3) and (d > 3e-5).mean() <= (0. if dtype == "float64" else 0.03)
        assert float(np.abs(ocf).max()) > 100.                                   # (the contacts act)
        # the constraint-space system of the inspect kernel (the same body-space arithmetic)
        tq, tdq = bw.to_device(Q, DQ, dt_)
        r = bw.inspect(tq, tdq, 5e-3, ["c_adm", "c_vel", "c_jac", "c_active", "c_frame", "dq_next"], cforce=bw.new_cforce(len(Q), dt_))
        torch.cuda.synchronize()
        tol = 1e-9 if dtype == "float64" else 2e-6
        act = np.asarray(dbg["active"]).astype(bool)
        assert np.array_equal(r["c_active"].cpu().numpy().astype(bool), act)
        for key, ok in (("c_adm", "adm"), ("c_vel", "vel0")):
            a = r[key].double().cpu().numpy().reshape(len(Q), -1); b = np.asarray(dbg[ok]).reshape(len(Q), -1)
            assert np.abs(a - b).max() < tol * np.abs(b).max(), (tag, key, np.abs(a - b).max() / np.abs(b).max())
        assert _rel(r["dq_next"].double().cpu().numpy(), res[False][1]).max() == 0.            # inspect == step, bit for bit
        bw.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_body_space_columns_of_the_four_contact_model(dtype):
    """Body-space columns on a model that fits one column set anyway (human36 with four contacts, the headline workload): the
    DEFAULT since round 6 (on request, ARB_STEP_BODY_COLUMNS, in round 5) -- against the oracle like the classical columns
    (ARB_STEP_CLASSIC_COLUMNS), and against them."""
    from conftest import assert_f32_parity
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g3_contacts.npz")
    m, _, _ = load_model("human36_c4")
    dt_ = getattr(torch, dtype)
    npt = np.float64 if dtype == "float64" else np.float32
    Q = np.concatenate([g["drop4_q"][:39], g["rand4_q"]]); DQ = np.concatenate([g["drop4_dq"][:39], g["rand4_dq"]])
    bw = BatchedWorlds(m)
    bc = 52 if dtype == "float32" else 20           # (float32: the kernels compiled for four contacts, FEAT bit 32)
    assert bw.plan(8192, 40, dtype=dt_)["feat"] == bc and bw.plan(8192, 40, dtype=dt_, body_columns=True)["feat"] == bc
    assert bw.plan(8192, 40, dtype=dt_, classic_columns=True)["feat"] == 4
    assert bw.plan(8192, 40, dtype=dt_, body_columns=True, general_kernels=True)["feat"] == 0
    oq, odq, ocf = O.step(m, Q.astype(npt).astype(np.float64), DQ.astype(npt).astype(np.float64), 5e-3)
    res = {}
    for bc in (False, True):
        tq, tdq = bw.to_device(Q, DQ, dt_)
        cf = bw.new_cforce(len(Q), dt_)
        bw.step(tq, tdq, 5e-3, 1, cforce=cf, classic_columns=not bc)
        torch.cuda.synchronize()
        res[bc] = (tq.double().cpu().numpy(), tdq.double().cpu().numpy())
        if dtype == "float64":
            assert _rel(res[bc][0], oq).max() < 1e-8 and _rel(res[bc][1], odq).max() < 1e-8
        else:
            assert_f32_parity(m, Q, DQ, 5e-3, res[bc][0], res[bc][1], oq, odq, 1e-5)
    d = np.maximum(_rel(res[False][0], res[True][0]), _rel(res[False][1], res[True][1]))
    assert d.max() > 0. and d.max() < (1e-9 if dtype == "float64" else 1e-3)
    r = bw.inspect(*bw.to_device(Q, DQ, dt_), 5e-3, ["dq_next"], cforce=bw.new_cforce(len(Q), dt_), body_columns=True)
    assert _rel(r["dq_next"].double().cpu().numpy(), res[True][1]).max() == 0.
    bw.close()


def test_body_space_columns_whole_episodes_and_torque_sequences():
    """Whole falling episodes of the eight-contact model on the body-space kernels: two- and three-wave builds, the work
    queue, one launch per step and user torques give the same bits (they are builds of one source), the states stay finite,
    and the float64 kernels stay within 1e-6 of the general float64 kernels over the 16 steps around the impact."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c8")
    bw = BatchedWorlds(m)
    B, T = 5000, 40
    q, dq = synth.standing_states(m, B, seed=12, drop=0.03, vel=0.1)
    tau = torch.as_tensor(_torque_sequence(m, T, B, seed=3), dtype=torch.float32, device=bw.device).contiguous()
    res = {}
    for mode, kw in (("episode", {}), ("two_waves", dict(waves=2)), ("per_step", {}), ("static", dict(static_worlds=True))):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        if mode == "per_step":
            for t in range(T):
                bw.step(tq, tdq, 5e-3, 1, cforce=cf, ext_gforce=tau[t].contiguous())
        else:
            bw.step(tq, tdq, 5e-3, T, cforce=cf, ext_gforce=tau, **kw)
        torch.cuda.synchronize()
        bw.status()
        res[mode] = (tq, tdq, cf)
    assert bool(torch.isfinite(res["episode"][0]).all()) and float(res["episode"][2][:, :, 3].max()) > 100.
    for mode in ("two_waves", "per_step", "static"):
        assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(res["episode"], res[mode])), mode
    q[:, 7] -= 0.01
    fin = {}
    for gk in (False, True):
        tq, tdq = bw.to_device(q[:600], dq[:600], torch.float64)
        cf = bw.new_cforce(600, torch.float64)
        bw.step(tq, tdq, 5e-3, 16, cforce=cf, general_kernels=gk)
        torch.cuda.synchronize()
        fin[gk] = (tq.cpu().numpy(), tdq.cpu().numpy())
    d = np.maximum(_rel(fin[False][0], fin[True][0]), _rel(fin[False][1], fin[True][1]))
    print("float64, 16 steps: body-space vs two column sets: max %.2e, median %.2e" % (d.max(), np.median(d)))
    assert np.median(d) < 1e-9 and (d > 1e-6).mean() < 0.02
    bw.close()


# ---------------------------------------------------------------------------
# a human36 outside the model class of the specialised kernels: with a PD controller (bench.py: model_classes)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pd_controlled_human36_on_the_floor_against_the_oracle(dtype):
    """human36 + 4 floor contacts + a ProportionalDerivativeController on its 36 hinge dofs (controllers.py:141-158: the
    constant part into the generalized force, dt kp + kd into the impedance) runs the GENERAL kernels -- the model is outside
    the class of the specialised ones.  States of the reference's own drop trajectory and random contact states: one step
    against the oracle (float64 1e-8; float32 the 1e-5 gate), and a 24-step episode in one launch against 24 one-step
    launches, bit for bit."""
    from conftest import assert_f32_parity
    from arboris_python_amd import scenes
    from arboris_python_amd.batch import BatchedWorlds
    m = scenes.flat(scenes.human36_world(contacts=4, pd=True))
    assert m.has_pd and m.nc == 4
    g = load_golden("g3_contacts.npz")
    Q = np.concatenate([g["drop4_q"][:39], g["rand4_q"]]); DQ = np.concatenate([g["drop4_dq"][:39], g["rand4_dq"]])
    dt_ = getattr(torch, dtype)
    npt = np.float64 if dtype == "float64" else np.float32
    bw = BatchedWorlds(m)
    p = bw.plan(8192, 40, dtype=dt_)
    assert p["feat"] == 0 and p["waves_per_simd"] == (3 if dtype == "float32" else 2), p
    qi, dqi = Q.astype(npt).astype(np.float64), DQ.astype(npt).astype(np.float64)
    oq, odq, ocf = O.step(m, qi, dqi, 5e-3)
    tq, tdq = bw.to_device(Q, DQ, dt_)
    cf = bw.new_cforce(len(Q), dt_)
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
    torch.cuda.synchronize()
    gq, gdq = tq.double().cpu().numpy(), tdq.double().cpu().numpy()
    if dtype == "float64":
        assert _rel(gq, oq).max() < 1e-8 and _rel(gdq, odq).max() < 1e-8
        assert np.abs(cf.cpu().numpy() - ocf).max() < 1e-7 * max(1., np.abs(ocf).max())
    else:
        assert_f32_parity(m, Q, DQ, 5e-3, gq, gdq, oq, odq, 1e-5)
    # the PD torques matter: without the controller the step differs
    m0 = scenes.flat(scenes.human36_world(contacts=4))
    o0 = O.step(m0, qi, dqi, 5e-3)
    assert _rel(o0[1], odq).max() > 1e-3
    res = {}
    for mode in ("episode", "per_step"):
        tq, tdq = bw.to_device(Q, DQ, dt_)
        cf = bw.new_cforce(len(Q), dt_)
        if mode == "episode":
            bw.step(tq, tdq, 5e-3, 24, cforce=cf)
        else:
            for _ in range(24):
                bw.step(tq, tdq, 5e-3, 1, cforce=cf)
        torch.cuda.synchronize()
        res[mode] = (tq, tdq, cf)
    assert all(torch.equal(a, b) for a, b in zip(res["episode"], res["per_step"]))
    assert torch.isfinite(res["episode"][0]).all()
    bw.close()
