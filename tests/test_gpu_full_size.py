"""BASELINE.json configurations at their FULL sizes on the GPU, checked through
size-independent properties (the CPU oracle cannot step 65536 worlds in test time):

  * replay: states logged by the device along a rollout are fed, world by world, to the float64
    oracle for ONE step and compared with the device's next logged state ("fp32 vs fp64 CPU tol
    check" of config 5) on a subsample of worlds and steps;
  * batch-position independence: a world's trajectory does not depend on where it sits in the
    batch or on the batch size (bitwise, same execution path);
  * launch-shape independence: N steps in one launch == N one-step launches (bitwise).

Tolerance: max|x_gpu - x_ref| / max(1, max|x_ref|) <= 1e-5 per world for one float32 step from
identical inputs (north star).  At least 99.5 % of the sampled world-steps must meet it (measured:
99.9 %, tools/replay_stats.py) and EVERY outlier must be explained: the float32 device and the
float64 oracle took different decisions in that step -- a different active set (constraints.py:292)
or a different release / static / sliding decision in some solve of the Gauss-Seidel sweeps
(constraints.py:781, 799; the device reports the decision of every solve, arb_inspect_out.gs_trace) -- AND that
decision was marginal for the oracle itself: at the first solve where the traces part, the oracle's inequality is
within 1e-5 of equality or the oracle's own trace changes under a one-ulp (float32) change of its input
or the oracle's decision at that very solve changes under a one-ulp change of the solve's own inputs
(tests/parity_tools.py; round 3: a decision difference alone no longer counts).  Errors with identical decisions are
accepted when the step is so ill-conditioned that the float64 oracle itself moves by at least the observed error when
its input moves by one float32 ulp (singular contact blocks: cond 1e18).  Unexplained outliers fail the test; explained ones stay
below 1e-3.
"""
import numpy as np
import pytest

import arb_oracle as O
from conftest import load_model, oracle_sensitivity
from parity_tools import check_replay, replay_errors, world_err

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

F32_TOL = 1e-5


@pytest.fixture(scope="module")
def bws():
    from arboris_python_amd.batch import BatchedWorlds
    cache = {}

    def get(name):
        if name not in cache:
            m, _, _ = load_model(name)
            cache[name] = (BatchedWorlds(m), m)
        return cache[name]
    yield get
    for bw, _ in cache.values():
        bw.close()


def _report(tag, r):
    print("%s replay: %d world-steps, ok %.4f, max err q %.2e dq %.2e, outliers by criterion %s, above the cap %d"
          % (tag, r["n"], r["ok"], r["max_q"], r["max_dq"], r["criteria"], len(r["over_cap"])))


# ---------------------------------------------------------------------------
# config 2: human36, no contacts, batch 1024, float32
# ---------------------------------------------------------------------------
def test_config2_human36_nocontact_1024(bws):
    from arboris_python_amd import synth
    bw, m = bws("human36_g")
    # (16 steps: the random joint velocities of this configuration, U(-3, 3) rad/s on a body without joint
    # limits or damping, grow without bound under the reference's integrator -- |dq| ~ 1e4 after 30 steps)
    B, T, dt = 1024, 16, 5e-3
    q, dq = synth.random_states(m, B, seed=0)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    log = bw.rollout(tq, tdq, dt, T, log_energy=False)
    torch.cuda.synchronize()
    assert torch.isfinite(log["q"]).all() and torch.isfinite(log["dq"]).all()
    worlds = np.arange(0, B, 64)
    eq, edq = replay_errors(m, log["q"], log["dq"], (0, 5, 10, 14), worlds, dt)
    assert eq.max() < F32_TOL and edq.max() < F32_TOL, (eq.max(), edq.max())
    # one launch of T steps == T launches of one step
    sq, sdq = bw.to_device(q, dq, torch.float32)
    for _ in range(T):
        bw.step(sq, sdq, dt, 1)
    torch.cuda.synchronize()
    assert torch.equal(sq, tq) and torch.equal(sdq, tdq)


# ---------------------------------------------------------------------------
# config 3: human36 falling on 4 floor contacts, batch 4096, 40 steps
# ---------------------------------------------------------------------------
def test_config3_falling_episode_4096(bws):
    from arboris_python_amd import synth
    bw, m = bws("human36_c4")
    B, T, dt = 4096, 40, 5e-3
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    log = bw.rollout(tq, tdq, dt, T, cforce=cf, log_energy=False)
    torch.cuda.synchronize()
    assert torch.isfinite(tq).all() and torch.isfinite(tdq).all()
    # the feet reached the floor and push on it
    assert float(cf[:, :, 3].max()) > 10.
    # replay of sampled (step, world) pairs through the oracle: free fall, first impacts, sliding
    worlds = np.arange(5, B, 64)                                    # 64 worlds x 13 steps = 832 world-steps
    _report("config 3", check_replay(bw, m, log, range(0, 39, 3), worlds, dt))
    # batch-position / batch-size independence over the whole episode, bitwise
    # (the library picks the float32 kernel build -- two or three waves per SIMD -- by batch size; the builds are
    # bit-identical, the pin below is belt and braces)
    sub = np.arange(3, B, 37)
    sq, sdq = bw.to_device(q[sub], dq[sub], torch.float32)
    bw.step(sq, sdq, dt, T, cforce=bw.new_cforce(len(sub), torch.float32), waves=3)
    torch.cuda.synchronize()
    assert torch.equal(sq, tq[sub]) and torch.equal(sdq, tdq[sub])
    # the other build of the kernel (two waves per SIMD; this batch runs the three-wave build) is bit-identical over the
    # whole episode: both execute the same float operations per world (-ffp-contract=on: no optimiser-dependent fusion)
    wq, wdq = bw.to_device(q, dq, torch.float32)
    bw.step(wq, wdq, dt, T, cforce=bw.new_cforce(B, torch.float32), waves=2)
    torch.cuda.synchronize()
    assert torch.equal(wq, tq) and torch.equal(wdq, tdq)
    # one launch of T steps == T launches of one step, bitwise
    pq, pdq = bw.to_device(q, dq, torch.float32)
    pcf = bw.new_cforce(B, torch.float32)
    for _ in range(T):
        bw.step(pq, pdq, dt, 1, cforce=pcf)
    torch.cuda.synchronize()
    assert torch.equal(pq, tq) and torch.equal(pdq, tdq) and torch.equal(pcf, cf)


# ---------------------------------------------------------------------------
# config 4: snake-64 (64 revolute joints), batch 16384, float64 kernels
# ---------------------------------------------------------------------------
def test_config4_snake64_16384(bws):
    from arboris_python_amd import synth
    bw, m = bws("snake64_g")
    B, dt = 16384, 1e-3
    q, dq = synth.random_states(m, B, seed=0, angle=0.5, vel=1.0)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    log = bw.rollout(tq, tdq, dt, 4, log_energy=False)
    torch.cuda.synchronize()
    assert torch.isfinite(tq).all() and torch.isfinite(tdq).all()
    worlds = np.arange(0, B, 256)                                   # 64 worlds x 3 steps (the oracle inverts 64x64 matrices)
    eq, edq = replay_errors(m, log["q"], log["dq"], (0, 1, 2), worlds, dt)
    # cond(Z) ~ 3e8: the reference's explicit inverse is itself only good to ~3e-6
    assert eq.max() < F32_TOL and edq.max() < F32_TOL, (eq.max(), edq.max())
    sub = np.arange(1, B, 1111)
    sq, sdq = bw.to_device(q[sub], dq[sub], torch.float64)
    bw.step(sq, sdq, dt, 4)
    torch.cuda.synchronize()
    assert torch.equal(sq, tq[sub]) and torch.equal(sdq, tdq[sub])


# ---------------------------------------------------------------------------
# config 5: human36 + contacts, 65536 worlds in flight, 32 steps; and the literal MPC shape
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["human36_c4", "human36_c8"])
def test_config5_65536_worlds_32_steps(bws, name):
    from arboris_python_amd import synth
    bw, m = bws(name)
    B, T, dt = 65536, 32, 5e-3
    q, dq = synth.standing_states(m, B, seed=5, drop=0.03, vel=0.1)
    q[:, 7] -= 0.02                                   # feet near the floor: contacts work from the first steps
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    log = bw.rollout(tq, tdq, dt, T, cforce=cf, log_energy=False)          # default execution: fused kernel
    torch.cuda.synchronize()
    assert torch.isfinite(tq).all() and torch.isfinite(tdq).all() and torch.isfinite(cf).all()
    worlds = np.arange(17, B, 1024)                                 # 64 worlds x 6 steps = 384 world-steps
    _report("config 5 (%s)" % name, check_replay(bw, m, log, (0, 5, 9, 16, 24, 30), worlds, dt))
    # batch-position independence at this size, first step, bitwise
    sub = np.arange(11, B, 997)
    sq, sdq = bw.to_device(q[sub], dq[sub], torch.float32)
    bw.step(sq, sdq, dt, 1, cforce=bw.new_cforce(len(sub), torch.float32), waves=3)      # (ignored by the 8-contact model: one build)
    torch.cuda.synchronize()
    assert torch.equal(sq, log["q"][1][sub]) and torch.equal(sdq, log["dq"][1][sub])


@pytest.mark.parametrize("name,stride", [("human36_c4", 16), ("human36_c8", 16)])
def test_replay_10k_world_steps_every_outlier_adjudicated(bws, name, stride):
    """The adjudication at scale, inside the suite the driver runs (round 4): every step of the 40-step episode x every
    16th world of the 4096-world headline batch = 256 x 39 = 9984 world-steps, with the 4 contacts of the headline and
    with the reference's own 8 (tests/test_human36_falling.py:32).  Measured (profiles/r04_replay_stats.txt): 99.9 % /
    99.6 % within 1e-5; every world-step above is adjudicated by parity_tools.explain_outlier, none unexplained; the
    world-steps above the cap (1e-3 in q, 1e-2 in dq) are decision differences that a criterion on the ORACLE's side
    explains, printed one by one."""
    from arboris_python_amd import synth
    bw, m = bws(name)
    B, T, dt = 4096, 40, 5e-3
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), log_energy=False)
    torch.cuda.synchronize()
    worlds = np.arange(3, B, stride)
    r = check_replay(bw, m, log, range(0, 39), worlds, dt, verbose=False)
    _report("%s, %d world-steps" % (name, r["n"]), r)
    assert r["n"] >= 9984


def test_config5_mpc_shape_against_the_oracle(bws):
    """The MPC shape against the ORACLE (round 4; until then the device was only compared with itself): 64 of the 2048
    rollouts x the 32 steps of the horizon, per-rollout user torques, replayed step by step through
    O.step(..., ext_gforce): >= 99.5 % of the 2048 world-steps within 1e-5, the others decision differences (below 1e-2;
    the torques are small against the contact forces, so the statistics are those of config 3)."""
    from arboris_python_amd import synth
    bw, m = bws("human36_c4")
    B, T, dt = 2048, 32, 5e-3
    q, dq = synth.standing_states(m, B, seed=9, drop=0.03, vel=0.1)
    rng = np.random.default_rng(9)
    tau_h = rng.uniform(-0.05, 0.05, size=(B, m.ndof))
    tau_h[:, :6] = 0.
    tau = torch.as_tensor(tau_h, dtype=torch.float32, device=bw.device).contiguous()
    tq, tdq = bw.to_device(q, dq, torch.float32)
    log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), ext_gforce=tau, log_energy=False)
    torch.cuda.synchronize()
    worlds = np.arange(7, B, 32)
    eq, edq = replay_errors(m, log["q"], log["dq"], range(0, T - 1), worlds, dt, ext=tau.double().cpu().numpy())
    ok = (eq < F32_TOL) & (edq < F32_TOL)
    print("MPC shape vs oracle: %d world-steps, ok %.4f, max err q %.2e dq %.2e" % (len(ok), ok.mean(), eq.max(), edq.max()))
    assert ok.mean() >= 0.995 and eq.max() < 1e-2 and edq.max() < 1e-1
    # the torques are in the oracle's step: without them it is 1e-4 off on the first step
    eq0, edq0 = replay_errors(m, log["q"], log["dq"], (0,), worlds, dt)
    assert edq0.max() > 10 * F32_TOL


def test_config5_mpc_2048_rollouts_x_32_step_horizon(bws):
    """The literal MPC shape: 2048 rollouts, the 32-step horizon resident in ONE launch, per-rollout
    user torques; equal (bitwise) to 32 one-step launches."""
    from arboris_python_amd import synth
    bw, m = bws("human36_c4")
    B, T, dt = 2048, 32, 5e-3
    q, dq = synth.standing_states(m, B, seed=9, drop=0.03, vel=0.1)
    rng = np.random.default_rng(9)
    tau = torch.as_tensor(rng.uniform(-0.05, 0.05, size=(B, m.ndof)), dtype=torch.float32, device=bw.device)
    tau[:, :6] = 0.                                    # no torque on the floating base (small elsewhere: the distal bodies are light)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    bw.step(tq, tdq, dt, T, cforce=cf, ext_gforce=tau)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(B, torch.float32)
    for _ in range(T):
        bw.step(sq, sdq, dt, 1, cforce=scf, ext_gforce=tau)
    torch.cuda.synchronize()
    assert torch.isfinite(tq).all()
    assert torch.equal(sq, tq) and torch.equal(sdq, tdq) and torch.equal(scf, cf)
    # the torques matter
    nq, ndq = bw.to_device(q, dq, torch.float32)
    bw.step(nq, ndq, dt, T, cforce=bw.new_cforce(B, torch.float32))
    torch.cuda.synchronize()
    assert float((ndq - tdq).abs().max()) > 1e-3
