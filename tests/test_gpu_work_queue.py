"""Multi-step launches of more worlds than the chip holds wavefronts draw (chunk of steps, world) items from a
device-side queue instead of giving every world a workgroup of its own (include/arbstep.h, ARB_STEP_STATIC_WORLDS):
same arithmetic, world by world and step by step, so the results must be bit-identical -- for every kernel variant
that takes the queue: both precisions, one and two register sets, per-step dt, per-world PD targets, the logs of
arb_rollout, warm-started constraint forces, ragged batch sizes, and against the oracle."""
import numpy as np
import pytest

import arb_oracle as O
from conftest import load_model

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bws():
    from arboris_python_amd.batch import BatchedWorlds
    cache = {}

    def get(name):
        if name not in cache:
            m, q0, dq0 = load_model(name)
            cache[name] = (BatchedWorlds(m), m, q0, dq0)
        return cache[name]
    yield get
    for bw, *_ in cache.values():
        bw.close()


def _states(m, B, seed):
    from arboris_python_amd import synth
    q, dq = synth.standing_states(m, B, seed=seed, drop=0.03, vel=0.1)
    q[:, 7] -= 0.02                                    # the feet reach the floor within the first steps
    return q, dq


@pytest.mark.parametrize("name,B,T", [("human36_c4", 4096 + 37, 24), ("human36_c4", 4096, 3), ("human36_c4", 5000, 5),
                                      ("human36_c4", 2049, 41), ("human36_c4", 2050, 7), ("human36_c4", 3001, 4),
                                      ("human36_c8", 4096, 10), ("human36_g", 8192 + 5, 9), ("human36_g", 8192, 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_queue_equals_static_bitwise(bws, name, B, T, dtype):
    bw, m, _, _ = bws(name)
    q, dq = _states(m, B, seed=11)
    out = {}
    for static in (True, False):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype) if m.nc else None
        bw.step(tq, tdq, 5e-3, T, cforce=cf, static_worlds=static)
        out[static] = (tq, tdq, cf)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out[False][0]).all())
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    if m.nc:
        assert float(out[False][2][:, :, 3].max()) > 10.            # contacts are working
        assert torch.equal(out[True][2], out[False][2])


@pytest.mark.parametrize("name", ["simplearm_pd", "txtytz", "jointlimits_min", "shapes_box_ball", "loop_arm", "snake9_free_g"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_queue_small_register_tiles(bws, name, dtype):
    """the 16- and 32-row kernels hold more wavefronts per CU: a batch large enough to take the queue there too"""
    bw, m, q0, dq0 = bws(name)
    B, T = 40000, 6
    rng = np.random.default_rng(3)
    q = np.repeat(q0[None], B, 0)
    dq = np.repeat(dq0[None], B, 0) + 0.05 * rng.standard_normal((B, m.ndof))
    out = {}
    for static in (True, False):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype) if m.nc else None
        bw.step(tq, tdq, 2e-3, T, cforce=cf, static_worlds=static)
        out[static] = (tq, tdq, cf)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out[False][0]).all())
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    if m.nc:
        assert torch.equal(out[True][2], out[False][2])


def test_queue_per_step_dt_and_logs(bws):
    """arb_rollout (logs indexed by the absolute step) and a non-uniform timeline through the queue"""
    bw, m, _, _ = bws("human36_c4")
    B, T = 4096, 13
    q, dq = _states(m, B, seed=5)
    dts = np.linspace(3e-3, 6e-3, T)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    log = bw.rollout(tq, tdq, dts, T, cforce=cf)
    # reference: one launch per step (no queue: a single step per launch)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(B, torch.float32)
    for k in range(T):
        assert torch.equal(log["q"][k], sq) and torch.equal(log["dq"][k], sdq), k
        bw.step(sq, sdq, float(dts[k]), 1, cforce=scf)
    torch.cuda.synchronize()
    assert torch.equal(sq, tq) and torch.equal(sdq, tdq) and torch.equal(scf, cf)


def test_queue_per_world_pd_targets(bws):
    bw, m, _, _ = bws("human36_c4")
    B, T = 4096, 8
    q, dq = _states(m, B, seed=6)
    rng = np.random.default_rng(0)
    n = m.ndof
    kp = torch.tensor(rng.uniform(5., 50., (B, n)), dtype=torch.float32, device="cuda")
    kd = torch.tensor(rng.uniform(.5, 5., (B, n)), dtype=torch.float32, device="cuda")
    qd = torch.tensor(rng.uniform(-.2, .2, (B, n)), dtype=torch.float32, device="cuda")
    dqd = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    out = {}
    for static in (True, False):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        bw.step(tq, tdq, 5e-3, T, cforce=cf, pd_targets=(qd, dqd), pd_gains=(kp, kd), static_worlds=static)
        out[static] = (tq, tdq, cf)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out[True], out[False]))


def test_queue_warm_started_forces(bws):
    """BallAndSocket forces persist from step to step (constraints.py:235-237): they travel between the chunks
    of a world through `cforce`; without a cforce buffer the launch must not take the queue (and stays correct)."""
    bw, m, q0, dq0 = bws("ballsocket")
    B, T = 8192, 12
    rng = np.random.default_rng(1)
    q = np.repeat(q0[None], B, 0)                      # (a FreeJoint's 16 pose entries: only the velocities are varied)
    dq = np.repeat(dq0[None], B, 0) + 0.1 * rng.standard_normal((B, m.ndof))
    out = {}
    for key, static, with_cf in (("static", True, True), ("queue", False, True), ("nocf", False, False)):
        tq, tdq = bw.to_device(q, dq, torch.float64)
        cf = bw.new_cforce(B, torch.float64) if with_cf else None
        bw.step(tq, tdq, 5e-3, T, cforce=cf, static_worlds=static)
        out[key] = (tq, tdq)
    torch.cuda.synchronize()
    for key in ("queue", "nocf"):
        assert torch.equal(out["static"][0], out[key][0]) and torch.equal(out["static"][1], out[key][1]), key
    # and the oracle on a few worlds
    ws = np.arange(0, B, B // 8)
    oq, odq, ocf = q[ws], dq[ws], None
    for _ in range(T):
        oq, odq, ocf = O.step(m, oq, odq, 5e-3, ocf)
    np.testing.assert_allclose(out["queue"][0][ws].cpu().numpy(), oq, rtol=0, atol=1e-8)
    np.testing.assert_allclose(out["queue"][1][ws].cpu().numpy(), odq, rtol=0, atol=1e-7)


@pytest.mark.parametrize("dtype,mixed", [(torch.float64, None), (torch.float32, False), (torch.float32, True)])
def test_queue_snake64_64_row_kernels(bws, dtype, mixed):
    """the 64-row kernels: float64 -- and the mixed build, whose register tile is float64 -- take the queue without the
    in-kernel item loop (csrc comment: one workgroup per item), plain float32 with it (float32 is not accurate on this model,
    cond(Z) ~ 3e8, but it is deterministic).  (float32 buffers are pinned to a build here: by default such a launch is
    promoted to the float64 kernels, which keep the state in float64 BETWEEN the steps of one launch -- T steps in one launch
    are then more accurate than T one-step launches, not bit-identical to them; tests/test_gpu_round6.py)"""
    bw, m, q0, dq0 = bws("snake64_g")
    B, T = 3000, 9
    rng = np.random.default_rng(2)
    q = rng.uniform(-1., 1., (B, m.nq)); dq = rng.uniform(-1., 1., (B, m.ndof))
    tq, tdq = bw.to_device(q, dq, dtype)
    bw.step(tq, tdq, 1e-3, T, mixed=mixed)
    sq, sdq = bw.to_device(q, dq, dtype)
    for _ in range(T):
        bw.step(sq, sdq, 1e-3, 1, mixed=mixed)
    torch.cuda.synchronize()
    assert torch.equal(tq, sq) and torch.equal(tdq, sdq)
