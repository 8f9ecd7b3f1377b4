"""Round-6 GPU tests (through the C ABI, ABI 8).

* the generic Controller plugin path: `arb_step_args.ext_impedance` (the dense impedance of user-defined controllers,
  core.py:327-339, 814-817) batched against the oracle and the reference-generated fixture g14, and through the object API
  (`core.World.update_controllers` polls the user's `update(dt)` on the host every step, `simulate` runs on the device);
* the MIXED build (ARB_STEP_MIXED: float32 state, float64 elimination), the library's default for models float32 cannot
  eliminate: BASELINE config 4 (snake-64) at 1e-5 through float32 buffers;
* `arb_inspect_ex`, the object API's flatten cache.
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


def _g14():
    from arboris_python_amd.flatten import FlatModel
    g = load_golden("g14_user_controller.npz")
    skip = ("q", "dq", "ctrl_gforce", "ctrl_impedance", "Z", "gforce0", "gforce", "cforce", "dt")
    return g, FlatModel.from_npz_dict({k: g[k] for k in g.files if k not in skip})


# ---------------------------------------------------------------------------
# ext_impedance, batched
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-8), ("float32", F32_TOL)])
def test_user_controller_fixture_of_the_reference_step_by_step(dtype, tol):
    """g14: the reference's own loop with a user-defined Controller (dense impedance) on human36 + four contacts.  Every one
    of the 12 recorded steps from the reference's state, fed the (gforce_a, Z_a) the reference's controller returned:
    impedance, controller force, constraint forces and the new state."""
    from arboris_python_amd.batch import BatchedWorlds
    g, m = _g14()
    dt, T = float(g["dt"]), len(g["ctrl_gforce"])
    tdt = getattr(torch, dtype)
    bw = BatchedWorlds(m)
    tq, tdq = bw.to_device(g["q"][:T], g["dq"][:T], tdt)          # the 12 steps as a batch of 12 worlds
    eg = torch.as_tensor(g["ctrl_gforce"], dtype=tdt, device=bw.device).contiguous()
    ez = torch.as_tensor(g["ctrl_impedance"], dtype=tdt, device=bw.device).contiguous()
    cf_in = np.zeros((T, m.nc, 4))
    cf = torch.as_tensor(cf_in, dtype=tdt, device=bw.device).contiguous()
    r = bw.inspect(tq, tdq, dt, ["Z", "gforce0", "gforce", "c_force", "q_next", "dq_next"], cforce=cf, ext_gforce=eg, ext_impedance=ez)
    Z = r["Z"].double().cpu().numpy()
    zs = np.abs(g["Z"]).max(axis=(1, 2), keepdims=True)
    assert (np.abs(Z - g["Z"]) / zs).max() < (1e-12 if dtype == "float64" else 1e-6)
    assert _rel(r["gforce0"].double().cpu().numpy(), g["gforce0"]).max() < (1e-10 if dtype == "float64" else 1e-5)
    bw.step(tq, tdq, dt, 1, cforce=cf, ext_gforce=eg, ext_impedance=ez)
    torch.cuda.synchronize()
    eq = _rel(tq.double().cpu().numpy(), g["q"][1:])
    edq = _rel(tdq.double().cpu().numpy(), g["dq"][1:])
    ef = np.abs(cf.double().cpu().numpy() - g["cforce"]).max() / np.abs(g["cforce"]).max()
    print("g14 %s: err q %.2e dq %.2e force %.2e" % (dtype, eq.max(), edq.max(), ef))
    if dtype == "float64":
        assert eq.max() < tol and edq.max() < tol and ef < 1e-7
    else:
        # float32 on impact steps (contact forces up to 1.3 kN on a 73 kg body): the gate of every float32 test -- 1e-5, or the
        # step is that ill-conditioned for the float64 oracle itself (conftest.assert_f32_parity: one float32 ulp on the input
        # moves the oracle by at least half the observed error; capped at 3e-5)
        from conftest import assert_f32_parity
        assert_f32_parity(m, g["q"][:T], g["dq"][:T], dt, tq.double().cpu().numpy(), tdq.double().cpu().numpy(), g["q"][1:], g["dq"][1:],
                          ext_gforce=g["ctrl_gforce"], ext_impedance=g["ctrl_impedance"])
        assert ef < 2e-4
    # the inspect kernel's next state is the step kernel's
    assert _rel(r["dq_next"].double().cpu().numpy(), tdq.double().cpu().numpy()).max() < (1e-12 if dtype == "float64" else 1e-6)
    # without the impedance the step is another step: the input is really used
    sq, sdq = bw.to_device(g["q"][:T], g["dq"][:T], tdt)
    bw.step(sq, sdq, dt, 1, cforce=torch.zeros_like(cf), ext_gforce=eg)
    torch.cuda.synchronize()
    assert _rel(sdq.double().cpu().numpy(), g["dq"][1:]).max() > 100 * tol
    bw.close()


@pytest.mark.parametrize("name,B,T", [("human36_c4", 300, 6), ("human36_g", 5000, 8), ("simplearm", 6000, 5), ("snake64_g", 40, 3)])
def test_random_dense_impedance_against_the_oracle(name, B, T):
    """A random dense (non-symmetric) impedance per world + user torques, T steps in ONE launch (through the work queue for
    the large batches; simplearm: one world per wavefront instead of the forest) against T oracle steps, float64; and the
    multi-step launch == T one-step launches bit for bit."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    rng = np.random.default_rng(61)
    n = m.ndof
    if m.nc:
        q, dq = synth.standing_states(m, B, seed=6, drop=0.01, vel=0.1)
    else:
        q, dq = synth.world_states(m, range(B), "random", 6, angle=0.5, vel=1.0)
    A = rng.uniform(-1., 1., size=(B, n, n))
    za = -(0.3 * A + 2.0 * np.eye(n)[None])                  # Z_a: damping-like (negative definite part) plus a dense non-symmetric part
    tau = rng.uniform(-0.2, 0.2, size=(B, n))
    dt = 2e-3
    tq, tdq = bw.to_device(q, dq, torch.float64)
    ez = torch.as_tensor(za, dtype=torch.float64, device=bw.device).contiguous()
    eg = torch.as_tensor(tau, dtype=torch.float64, device=bw.device).contiguous()
    cf = bw.new_cforce(B, torch.float64) if m.nc else None
    bw.step(tq, tdq, dt, T, cforce=cf, ext_gforce=eg, ext_impedance=ez)
    sq, sdq = bw.to_device(q, dq, torch.float64)
    scf = bw.new_cforce(B, torch.float64) if m.nc else None
    for _ in range(T):
        bw.step(sq, sdq, dt, 1, cforce=scf, ext_gforce=eg, ext_impedance=ez)
    torch.cuda.synchronize()
    assert torch.equal(tq, sq) and torch.equal(tdq, sdq)
    sub = np.arange(0, B, max(1, B // 24))
    oq, odq, ocf = q[sub], dq[sub], None
    for _ in range(T):
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf, ext_gforce=tau[sub], ext_impedance=za[sub])
    eq, edq = _rel(tq.cpu().numpy()[sub], oq), _rel(tdq.cpu().numpy()[sub], odq)
    print("%s: dense impedance, %d worlds x %d steps: err q %.2e dq %.2e" % (name, len(sub), T, eq.max(), edq.max()))
    # (contacts: T steps of a contact trajectory amplify the last bits; snake-64: cond(Z) ~ 3e8, the oracle's explicit inverse
    # is itself good to ~3e-6 only, tests/test_gpu_full_size.py)
    tol = 1e-6 if m.nc else 1e-5 if name == "snake64_g" else 1e-8
    assert eq.max() < tol and edq.max() < tol
    assert bw.plan(B, T, dtype=torch.float64, other_inputs=True)["worlds_per_wavefront"] >= 1
    bw.close()


def test_ext_impedance_argument_checks():
    from arboris_python_amd.batch import BatchedWorlds
    m, q0, dq0 = load_model("simplearm")
    bw = BatchedWorlds(m)
    tq, tdq = bw.to_device(q0[None], dq0[None], torch.float32)
    with pytest.raises(ValueError):
        bw.step(tq, tdq, 1e-3, 1, ext_impedance=torch.zeros((1, 3), device=bw.device))
    with pytest.raises(ValueError):
        bw.step(tq, tdq, 1e-3, 1, ext_impedance=torch.zeros((1, 3, 3), dtype=torch.float64, device=bw.device))
    with pytest.raises(ValueError):          # a torque sequence where one row is expected (was a TypeError from int(None))
        bw.inspect(tq, tdq, 1e-3, ["Z"], ext_gforce=torch.zeros((2, 1, 3), device=bw.device))
    a = _capi.StepArgs()
    a.q, a.dq, a.nworlds, a.nsteps, a.dt = tq.data_ptr(), tdq.data_ptr(), 1, 1, 1e-3
    out = _capi.InspectOut()
    a.ext_gforce_steps = tq.data_ptr()
    assert bw._lib.arb_inspect_ex(bw._handle, _capi.ARB_F32, a, out, None) == 1       # sequences: not in inspect
    assert bw._lib.arb_inspect_ex(bw._handle, _capi.ARB_F32, None, out, None) == 1
    # plan() knows about the cost: a small model with a running cost runs one world per wavefront
    big = 100000
    assert bw.plan(big, 8)["worlds_per_wavefront"] == bw.info["forest_copies"] > 1
    assert bw.plan(big, 8, cost=True)["worlds_per_wavefront"] == 1
    bw.close()


# ---------------------------------------------------------------------------
# the plugin path through the object API
# ---------------------------------------------------------------------------
def _own_world_with_user_controller():
    from arboris_python_amd.core import World, Controller
    from arboris_python_amd.robots.human36 import add_human36
    from arboris_python_amd.robots.simpleshapes import add_groundplane
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.constraints import get_all_contacts
    from plugin_controllers import make_spring_damper
    FOUR = ('Right foot toe tip', 'Right foot heel', 'Left foot toe tip', 'Left foot heel')
    w = World()
    add_groundplane(w)
    add_human36(w)
    w.register(WeightController())
    ctrl = make_spring_damper(Controller)()
    w.register(ctrl)
    for c in get_all_contacts(w, friction_coeff=.6):
        if c._shapes[1].name in FOUR:
            w.register(c)
    w.init()
    return w, ctrl


def test_user_defined_controller_runs_simulate_on_the_device_and_matches_the_reference():
    """The SAME controller class the reference ran for g14 (tests/plugin_controllers.py), derived from THIS package's
    Controller, registered on this package's human36: `simulate`'s loop body on the device, 12 steps, against the
    reference's trajectory, impedance and forces (float64, 1e-8)."""
    from arboris_python_amd.flatten import JT_FREE
    g, m = _g14()
    w, ctrl = _own_world_with_user_controller()
    for b, j in enumerate(w.iterjoints()):
        qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
        if m.jtype[b] == JT_FREE:
            j.gpos = g["q"][0][qs].reshape(4, 4).copy()
        else:
            j.gpos[:] = g["q"][0][qs]
    w._gvel[:] = g["dq"][0]
    dt = float(g["dt"])
    cons = list(w._constraints)
    for k in range(len(g["ctrl_gforce"])):
        w.update_dynamic()
        w.update_controllers(dt)
        zs = np.abs(g["Z"][k]).max()
        assert np.abs(w._impedance - g["Z"][k]).max() / zs < 1e-11, k
        assert _rel(w._gforce, g["gforce0"][k]) < 1e-9, k
        assert np.abs(w._admittance @ w._impedance - np.eye(w.ndof)).max() < 1e-8
        w.update_constraints(dt)
        assert _rel(w._gforce, g["gforce"][k]) < 1e-7, k
        f = np.array([c._force for c in cons])
        assert np.abs(f - g["cforce"][k]).max() <= 1e-7 * max(1., np.abs(g["cforce"][k]).max()), k
        w.integrate(dt)
        q = np.concatenate([np.asarray(j.gpos, float).ravel() for j in w.iterjoints()])
        assert _rel(q, g["q"][k + 1]) < 1e-8 and _rel(w._gvel, g["dq"][k + 1]) < 1e-8, k
    # the model was flattened ONCE for the 48 stage calls (round 6: cached on a signature of everything but the state)
    assert w._engine.flatten_count == 1
    # ... and again as soon as something flatten_world reads changes
    ctrl_pd_free = w._engine.flatten_count
    cons[0].disable()
    w.update_dynamic()
    assert w._engine.flatten_count == ctrl_pd_free + 1


def test_batched_worlds_refuses_a_world_with_a_user_controller_and_says_what_to_do():
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd.flatten import UnsupportedModelError
    w, _ = _own_world_with_user_controller()
    with pytest.raises(UnsupportedModelError, match="ext_impedance"):
        BatchedWorlds(w)


# ---------------------------------------------------------------------------
# the mixed build: BASELINE config 4 through float32 buffers
# ---------------------------------------------------------------------------
def test_config4_snake64_through_float32_buffers():
    """snake-64 x 2048 (one GPU's shard of BASELINE config 4), float32 buffers, 4 steps in one launch.  By default the launch
    is PROMOTED to the float64 kernels (the plan is the float64 plan) and every replayed world-step is within 1e-5 of the
    float64 oracle; the mixed build on request (float32 LDS footprint, more wave slots, 1.28 x the throughput) is within
    2e-4; `mixed=False` on the same states is wrong by orders of magnitude and warns."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    from parity_tools import replay_errors
    m, _, _ = load_model("snake64_g")
    bw = BatchedWorlds(m)
    B, dt, T = 2048, 1e-3, 4
    q, dq = synth.random_states(m, B, seed=0, angle=0.5, vel=1.0)
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    worlds = np.arange(0, B, 32)                                   # 64 worlds (the oracle inverts 64x64 matrices)
    # default: promoted.  One step against the oracle ...
    tq, tdq = bw.to_device(q, dq, torch.float32)
    bw.step(tq, tdq, dt, 1)
    torch.cuda.synchronize()
    assert bw.warnings() == 0
    o_q, o_dq, _ = O.step(m, f32(q[worlds]), f32(dq[worlds]), dt)
    eq, edq = _rel(tq.double().cpu().numpy()[worlds], o_q), _rel(tdq.double().cpu().numpy()[worlds], o_dq)
    print("snake-64 float32 buffers, default (promoted): %d worlds, max err q %.2e dq %.2e" % (len(eq), eq.max(), edq.max()))
    assert eq.max() < F32_TOL and edq.max() < F32_TOL
    # ... and T steps in one launch == the float64 kernels on the same (float32-representable) states, rounded once at the end
    tq, tdq = bw.to_device(q, dq, torch.float32)
    dq64, ddq64 = bw.to_device(f32(q), f32(dq), torch.float64)
    bw.step(tq, tdq, dt, T)
    bw.step(dq64, ddq64, dt, T)
    torch.cuda.synchronize()
    assert torch.equal(tq, dq64.float()) and torch.equal(tdq, ddq64.float())
    p = bw.plan(B, T, dtype=torch.float32)
    pm = bw.plan(B, T, dtype=torch.float32, mixed=True)
    pf = bw.plan(B, T, dtype=torch.float32, mixed=False)
    pd = bw.plan(B, T, dtype=torch.float64)
    print("plans: default %s\n       mixed %s\n       float32 %s\n       float64 %s" % (p, pm, pf, pd))
    assert p == pd and bw.info["mixed_default"] == 2
    assert pm["lds_bytes"] == pf["lds_bytes"] < pd["lds_bytes"] and pm["wave_slots"] > pd["wave_slots"]
    # the mixed build on request
    mq, mdq = bw.to_device(q, dq, torch.float32)
    mlog = bw.rollout(mq, mdq, dt, T, log_energy=False, mixed=True)
    torch.cuda.synchronize()
    assert bw.warnings() == 0
    meq, medq = replay_errors(m, mlog["q"], mlog["dq"], (0, 1, 2), worlds, dt)
    print("snake-64 float32 buffers, mixed build on request: max err q %.2e dq %.2e (median dq %.1e)" % (meq.max(), medq.max(), np.median(medq)))
    assert meq.max() < 2e-5 and medq.max() < 2e-4 and np.median(medq) < 2e-5
    # plain float32 on the same states
    sq, sdq = bw.to_device(q, dq, torch.float32)
    bw.step(sq, sdq, dt, 1, mixed=False)
    torch.cuda.synchronize()
    assert bw.warnings() == _capi.ARB_WARN_ILLCOND
    o_q, o_dq, _ = O.step(m, np.asarray(q[worlds], np.float32).astype(np.float64), np.asarray(dq[worlds], np.float32).astype(np.float64), dt)
    assert _rel(sdq.double().cpu().numpy()[worlds], o_dq).max() > 1e-3
    # the mixed build on request for a model that does not need it: human36 on the floor, equal to the default to float32 rounding
    mh, _, _ = load_model("human36_c4")
    bh = BatchedWorlds(mh)
    qh, dqh = synth.standing_states(mh, 256, seed=2, drop=0.0, vel=0.1)
    a_q, a_dq = bh.to_device(qh, dqh, torch.float32)
    b_q, b_dq = bh.to_device(qh, dqh, torch.float32)
    bh.step(a_q, a_dq, 5e-3, 1, cforce=bh.new_cforce(256, torch.float32))
    bh.step(b_q, b_dq, 5e-3, 1, cforce=bh.new_cforce(256, torch.float32), mixed=True)
    torch.cuda.synchronize()
    oq, odq, _ = O.step(mh, np.asarray(qh, np.float32).astype(np.float64), np.asarray(dqh, np.float32).astype(np.float64), 5e-3)
    ea, eb = _rel(a_dq.double().cpu().numpy(), odq), _rel(b_dq.double().cpu().numpy(), odq)
    print("human36_c4 one step: default err %.2e (median %.1e), mixed err %.2e (median %.1e)" % (ea.max(), np.median(ea), eb.max(), np.median(eb)))
    assert np.median(eb) <= np.median(ea) * 1.5 and eb.max() < 1e-4
    bh.close()
    bw.close()


# ---------------------------------------------------------------------------
# the single-process multi-device driver
# ---------------------------------------------------------------------------
def test_sharded_worlds_two_shards_on_one_device_equal_the_unsharded_launch_bitwise():
    """dist.ShardedWorlds, the library-level driver (one host thread, one handle and one stream per shard): with
    `devices=[0, 0, 0]` the three shards of a ragged batch run on this box's one GPU, each on its own stream, per-shard
    torque sequences and costs; gathered in world order they are the unsharded launch bit for bit."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd.dist import ShardedWorlds
    m, _, _ = load_model("human36_c4")
    B, T, dt = 1000, 12, 5e-3
    q, dq = synth.standing_states(m, B, seed=4, drop=0.02, vel=0.1)
    rng = np.random.default_rng(4)
    tau = rng.uniform(-0.05, 0.05, size=(T, B, m.ndof)); tau[:, :, :6] = 0.
    sw = ShardedWorlds(m, devices=[0, 0, 0])
    shards = sw.scatter(q, dq, torch.float32, cforce=True)
    assert [(s["lo"], s["hi"]) for s in shards] == [(0, 334), (334, 668), (668, 1000)]
    costs, per = [], []
    for s, sh in zip(sw.steppers, shards):
        c = dict(out=torch.zeros(sh["hi"] - sh["lo"], dtype=torch.float32, device=s.device),
                 w_dq=torch.full((m.ndof,), 0.1, dtype=torch.float32, device=s.device))
        costs.append(c)
        per.append(dict(cost=c, ext_gforce=torch.as_tensor(tau[:, sh["lo"]:sh["hi"]], dtype=torch.float32, device=s.device).contiguous()))
    sw.step(shards, dt, T, per_shard=per)
    out = sw.gather(shards, keys=("q", "dq", "cforce"), extra=[dict(cost=c["out"]) for c in costs])
    bw = BatchedWorlds(m)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    cost = dict(out=torch.zeros(B, dtype=torch.float32, device=bw.device), w_dq=torch.full((m.ndof,), 0.1, dtype=torch.float32, device=bw.device))
    bw.step(tq, tdq, dt, T, cforce=cf, ext_gforce=torch.as_tensor(tau, dtype=torch.float32, device=bw.device).contiguous(), cost=cost)
    torch.cuda.synchronize()
    assert torch.equal(out["q"], tq.cpu()) and torch.equal(out["dq"], tdq.cpu()) and torch.equal(out["cforce"], cf.cpu())
    assert torch.equal(out["cost"], cost["out"].cpu()) and float(out["cost"].min()) > 0.
    assert len({id(st) for st in sw.streams}) == 3
    sw.close(); bw.close()


# ---------------------------------------------------------------------------
# body-space columns as the default: any number of plane / sphere contacts
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("ncon", [1, 2, 3, 5, 7])
def test_body_space_columns_by_default_for_any_number_of_contacts(ncon):
    """Since round 6 every model of the class (enabled plane / sphere SoftFingerContacts only, no PD controller, no viscosity,
    one small tree) runs body-space constraint columns by default -- not only the 4- and 8-contact models of the benchmark:
    human36 with the first `ncon` of its eight foot points, float64 against the oracle (1e-8) over a short drop, the default
    against the classical columns (equal to rounding), float32 through `assert_f32_parity`."""
    from conftest import assert_f32_parity
    from arboris_python_amd.core import World
    from arboris_python_amd.robots.human36 import add_human36
    from arboris_python_amd.robots.simpleshapes import add_groundplane
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.constraints import get_all_contacts
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd import synth
    w = World()
    add_groundplane(w); add_human36(w); w.register(WeightController())
    for i, c in enumerate(get_all_contacts(w, friction_coeff=.6)):
        if i < ncon:
            w.register(c)
    w.init()
    m, _, _ = flatten_world(w)
    assert m.nc == ncon
    bw = BatchedWorlds(m)
    assert bw.plan(512, 8, dtype=torch.float64)["feat"] & 16 and bw.plan(8192, 40)["feat"] & 16
    assert not (bw.plan(8192, 40, classic_columns=True)["feat"] & 16)
    B, dt = 96, 5e-3
    q, dq = synth.standing_states(m, B, seed=60 + ncon, drop=0.01, vel=0.3)
    q[:, 7] -= 0.004
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cq, cdq = bw.to_device(q, dq, torch.float64)
    cf, ccf = bw.new_cforce(B, torch.float64), bw.new_cforce(B, torch.float64)
    oq, odq, ocf = q, dq, None
    for k in range(5):
        bw.step(tq, tdq, dt, 1, cforce=cf)
        bw.step(cq, cdq, dt, 1, cforce=ccf, classic_columns=True)
        oq, odq, ocf = O.step(m, oq, odq, dt, cforce=ocf)
    torch.cuda.synchronize()
    assert _rel(tq.cpu().numpy(), oq).max() < 1e-8 and _rel(tdq.cpu().numpy(), odq).max() < 1e-7
    assert _rel(cq.cpu().numpy(), oq).max() < 1e-8 and _rel(cdq.cpu().numpy(), odq).max() < 1e-7
    assert float(cf.abs().max()) > 10.
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    bw.step(sq, sdq, dt, 1, cforce=bw.new_cforce(B, torch.float32))
    torch.cuda.synchronize()
    rq, rdq, _ = O.step(m, f32(q), f32(dq), dt)
    assert_f32_parity(m, q, dq, dt, sq.double().cpu().numpy(), sdq.double().cpu().numpy(), rq, rdq)
    bw.close()
