"""Round-2 execution variants of the step (all through the C ABI):

  * ARB_STEP_SPLIT_WAVE: the Gauss-Seidel sweeps (core.py:929-935) in their own kernel, one wavefront per world,
    fed through HBM -- the same device code as the fused kernel's sweeps, so the results are bit-identical;
  * arb_step_args.dt_steps: a non-uniform timeline (core.py:1357, dt = next_time - current_time) inside ONE
    launch equals one launch per step with that step's dt, and matches the oracle's rollout;
  * arb_inspect_out.gs_trace: the decision of every local solve, against the oracle's trace.
"""
import numpy as np
import pytest

import arb_oracle as O
from conftest import load_golden, load_model, assert_f32_parity

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


@pytest.mark.parametrize("name,B", [("human36_c4", 4096), ("human36_c8", 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_wave_split_equals_fused_bitwise(bws, name, B, dtype):
    from arboris_python_amd import synth
    bw, m, _, _ = bws(name)
    if dtype == torch.float64:
        B //= 4
    q, dq = synth.standing_states(m, B, seed=3, drop=0.03, vel=0.1)
    q[:, 7] -= 0.015                                   # the feet reach the floor within the first steps
    T = 12
    fq, fdq = bw.to_device(q, dq, dtype)
    fcf = bw.new_cforce(B, dtype)
    # (the split execution runs the general kernels; a model with body-space constraint columns -- human36 with eight contacts,
    # round 5 -- steps through them by default, equal to rounding only: the fused reference is the general kernel too)
    bw.step(fq, fdq, 5e-3, T, cforce=fcf, general_kernels=True)
    sq, sdq = bw.to_device(q, dq, dtype)
    scf = bw.new_cforce(B, dtype)
    bw.step(sq, sdq, 5e-3, T, cforce=scf, split="wave")
    torch.cuda.synchronize()
    assert float(fcf[:, :, 3].max()) > 10.            # contacts are working
    assert torch.equal(sq, fq) and torch.equal(sdq, fdq) and torch.equal(scf, fcf)


def test_wave_split_ball_and_socket_and_joint_limits(bws):
    """the other constraint types through the split execution (warm-started forces, constraints.py:235-237)"""
    g = load_golden("g6_constraints.npz")
    bw, m, q0, dq0 = bws("ballsocket")
    tq, tdq = bw.to_device(q0[None], dq0[None], torch.float64)
    tcf = bw.new_cforce(1, torch.float64)
    for k in range(5):
        bw.step(tq, tdq, 0.001, 1, cforce=tcf, split="wave")
        torch.cuda.synchronize()
        assert np.abs(tcf.cpu().numpy()[0, 0, :3] - g["bs_force"][k]).max() < 1e-7
    bw, m, q0, dq0 = bws("jointlimits_max")
    tq, tdq = bw.to_device(q0[None], dq0[None], torch.float64)
    bw.step(tq, tdq, 1e-3, 99, cforce=bw.new_cforce(1, torch.float64), split="wave")
    torch.cuda.synchronize()
    assert np.abs(tq.cpu().numpy()[0] - g["jl_max_q"][99]).max() < 1e-8


@pytest.mark.parametrize("split", [False, "wave"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_non_uniform_timeline_in_one_launch(bws, dtype, tol, split):
    """simulate() takes dt from the timeline (core.py:1356-1357); here 24 steps with 24 different dt."""
    from arboris_python_amd import synth
    bw, m, _, _ = bws("human36_c4")
    B = 64
    rng = np.random.default_rng(12)
    dts = rng.uniform(2e-3, 6e-3, size=24)
    q, dq = synth.standing_states(m, B, seed=12, drop=0.02, vel=0.1)
    aq, adq = bw.to_device(q, dq, dtype)
    acf = bw.new_cforce(B, dtype)
    bw.step(aq, adq, dts, len(dts), cforce=acf, split=split)           # one call, the whole timeline
    bq, bdq = bw.to_device(q, dq, dtype)
    bcf = bw.new_cforce(B, dtype)
    for dt in dts:
        bw.step(bq, bdq, float(dt), 1, cforce=bcf, split=split)        # one call per step
    torch.cuda.synchronize()
    assert torch.equal(aq, bq) and torch.equal(adq, bdq) and torch.equal(acf, bcf)
    if split is False:
        # against the oracle's rollout (8 worlds; float32: error accumulated over 24 steps of a contact scenario)
        oq, odq, _ = O.rollout(m, q[:8].astype(np.float32 if dtype == torch.float32 else np.float64).astype(np.float64),
                               dq[:8].astype(np.float32 if dtype == torch.float32 else np.float64).astype(np.float64), dts)
        eq = np.abs(aq[:8].cpu().numpy() - oq).max(axis=1) / np.maximum(1., np.abs(oq).max(axis=1))
        edq = np.abs(adq[:8].cpu().numpy() - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
        assert eq.max() < tol * 5 and np.median(edq) < tol * 50, (eq.max(), edq.max())
    # the per-step logs of a rollout honour the timeline too
    if split is False and dtype == torch.float64:
        cq, cdq = bw.to_device(q, dq, dtype)
        log = bw.rollout(cq, cdq, dts, len(dts), cforce=bw.new_cforce(B, dtype), log_energy=False)
        torch.cuda.synchronize()
        assert torch.equal(cq, aq)
        o1q, o1dq, _ = O.step(m, log["q"][5][:4].cpu().numpy(), log["dq"][5][:4].cpu().numpy(), float(dts[5]))
        assert np.abs(log["q"][6][:4].cpu().numpy() - o1q).max() < 1e-9
    with pytest.raises(ValueError):
        bw.step(aq, adq, dts[:3], 24, cforce=acf)


def test_gs_trace_matches_oracle_decisions(bws):
    g = load_golden("g3_contacts.npz")
    bw, m, _, _ = bws("human36_c8")
    Q, DQ = g["drop8_q"][:39], g["drop8_dq"][:39]
    tq, tdq = bw.to_device(Q, DQ, torch.float64)
    r = bw.inspect(tq, tdq, 5e-3, ["gs_trace", "gs_stats"])
    dtr = r["gs_trace"].cpu().numpy()
    st = r["gs_stats"].cpu().numpy()
    tr = []
    O.step(m, Q, DQ, 5e-3, trace=tr)
    assert len(tr) > 1000
    seen = np.zeros(3, int)
    for t in tr:
        w, k, c = t["world"], t["sweep"], t["c"]
        if k < st[w, 4]:
            assert min(int(dtr[w, k, c]), 2) == t["branch"], (w, k, c, dtr[w, k, c], t["branch"])
            seen[t["branch"]] += 1
    assert (seen > 0).all(), seen
    # executed solves and nothing else are recorded
    for w in range(len(Q)):
        assert (dtr[w, st[w, 4]:] == -1).all()
        assert int((dtr[w] >= 0).sum()) == int(st[w, :4].sum())


@pytest.mark.parametrize("name", ["human36_g", "human36_c4", "human36_c8", "simplearm_g"])
def test_matrix_core_elimination_parity(bws, name):
    """ARB_STEP_MFMA_ELIM: phase C on v_mfma_f32_4x4x1 instead of the vector ALU (replaces numpy.linalg.inv,
    core.py:818, 925-927) -- float32 single steps against the oracle at the same 1e-5 gate as the default path,
    on the golden states of configs 1-3 (incl. the 75-column system of human36 + 8 contacts: two column sets)."""
    bw, m, q0, dq0 = bws(name)
    if name == "human36_g":
        g = load_golden("g2_human36.npz"); Q, DQ, dt = g["q"][g["dt"] == 5e-3], g["dq"][g["dt"] == 5e-3], 5e-3
    elif name == "simplearm_g":
        g = load_golden("g1_simplearm.npz"); Q, DQ, dt = g["traj_q"], g["traj_dq"], 0.01
    else:
        g = load_golden("g3_contacts.npz"); nc = m.nc
        Q = np.concatenate([g["drop%d_q" % nc][:39], g["rand%d_q" % nc]]); DQ = np.concatenate([g["drop%d_dq" % nc][:39], g["rand%d_dq" % nc]]); dt = 5e-3
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    oq, odq, _ = O.step(m, f(Q), f(DQ), dt)
    res = {}
    for mf in (False, True):
        tq, tdq = bw.to_device(Q, DQ, torch.float32)
        cf = bw.new_cforce(len(Q), torch.float32) if m.nc else None
        bw.step(tq, tdq, dt, 1, cforce=cf, mfma=mf)
        torch.cuda.synchronize()
        # (same gate as the default path; the ill-conditioned random contact states are judged as there)
        assert_f32_parity(m, Q, DQ, dt, tq.cpu().numpy(), tdq.cpu().numpy(), oq, odq, 1e-5)
        res[mf] = (tq, tdq)
    # exact float32 FMAs in the same pivot order: the two eliminations agree far below the gate
    d = (res[True][1] - res[False][1]).abs().max().item()
    assert d < 1e-4 * max(1., res[False][1].abs().max().item())
    # a 20-step launch stays finite and close to the default path
    tq, tdq = bw.to_device(Q[:4], DQ[:4], torch.float32)
    sq, sdq = bw.to_device(Q[:4], DQ[:4], torch.float32)
    cf1 = bw.new_cforce(4, torch.float32) if m.nc else None
    cf2 = bw.new_cforce(4, torch.float32) if m.nc else None
    bw.step(tq, tdq, dt, 20, cforce=cf1, mfma=True)
    bw.step(sq, sdq, dt, 20, cforce=cf2)
    torch.cuda.synchronize()
    assert torch.isfinite(tq).all() and torch.isfinite(tdq).all()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_split_executions_on_rank_deficient_blocks(bws, dtype):
    """The closed-loop planar arm (rank-2 3x3 admittance block, numpy.linalg.pinv semantics, constraints.py:235)
    through the split execution: the wave-per-world sweep kernel runs the fused kernel's code (bitwise equal)."""
    g = load_golden("g12_singular.npz")
    bw, m, _, _ = bws("loop_arm")
    Q, DQ, F = g["loop_q"], g["loop_dq"], g["loop_force"]
    cf0 = np.zeros((40, 1, 4)); cf0[1:, 0, :3] = F[:39]
    out = {}
    for split in (False, "wave"):
        tq, tdq = bw.to_device(Q[:40], DQ[:40], dtype)
        tcf = torch.as_tensor(cf0, dtype=dtype, device=bw.device).contiguous()
        bw.step(tq, tdq, 5e-3, 1, cforce=tcf, split=split)
        torch.cuda.synchronize()
        assert torch.isfinite(tq).all() and torch.isfinite(tcf).all()
        out[split] = (tq, tdq, tcf)
    assert all(torch.equal(a, b) for a, b in zip(out[False], out["wave"]))
    # and the forces are the reference's
    ftol = 1e-6 if dtype == torch.float64 else 5e-3
    assert np.abs(out["wave"][2].cpu().numpy()[:, 0, :3] - F).max() < ftol * max(1., np.abs(F).max())
