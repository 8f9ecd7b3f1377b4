"""Small worlds share wavefronts (the FOREST of a model of at most 16 dofs: arb_model_create builds k independent copies
of it as one model, and a large batch runs as nw / k forest worlds on the same buffers; include/arbstep.h,
ARB_STEP_ONE_WORLD; SURVEY 7.1 step 7: "pack several small worlds per wavefront for n << 64").

The forest launch is checked against the float64 reference (golden trajectories; float32: the oracle on the rounded
inputs) at the gates of the one-world kernels, and against the one-world launch of the same batch: copies share nothing
but ground, gravity and dt, the augmented system is block diagonal and products with its exact zeros change nothing.
"""
import numpy as np
import pytest

import arb_oracle as O
from conftest import load_golden, load_model
from test_gpu_parity import rel, F32_TOL

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


CASES = {
    # name: (golden file, q key, dq key, dt)
    "jointlimits_min": ("g6_constraints.npz", "jl_min_q", "jl_min_dq", 1e-3),
    "jointlimits_max": ("g6_constraints.npz", "jl_max_q", "jl_max_dq", 1e-3),
    "shapes_plane_ball": ("g7_shapes.npz", "plane_ball_q", "plane_ball_dq", 5e-3),
    "shapes_box_ball": ("g7_shapes.npz", "box_ball_q", "box_ball_dq", 5e-3),
    "shapes_ball_ball": ("g7_shapes.npz", "ball_ball_q", "ball_ball_dq", 5e-3),
    "shapes_dome_point": ("g7_shapes.npz", "dome_point_q", "dome_point_dq", 5e-3),
    "txtytz": ("g11_txtytz.npz", "roll_q", "roll_dq", 5e-3),
}


def _tiled(Q, DQ, B):
    reps = -(-B // (len(Q) - 1))
    q = np.tile(Q[:-1], (reps, 1))[:B]
    dq = np.tile(DQ[:-1], (reps, 1))[:B]
    qn = np.tile(Q[1:], (reps, 1))[:B]
    dqn = np.tile(DQ[1:], (reps, 1))[:B]
    return q, dq, qn, dqn


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", sorted(CASES))
def test_forest_single_steps_match_the_reference(name, dtype):
    """Every step of the reference's trajectory from the reference's own state, tiled to a batch that runs on the forest
    (more worlds than twice the wave slots, not a multiple of the copies: the last worlds run one per wavefront)."""
    from arboris_python_amd.batch import BatchedWorlds
    gfile, kq, kdq, dt = CASES[name]
    g = load_golden(gfile)
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    K = bw.info["forest_copies"]
    assert K >= 2, "a %d-dof model should have a forest" % m.ndof
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    B = 16 * cus + 3 * K + 1
    B += 1 if B % K == 0 else 0
    p = bw.plan(B, 1, dtype=dtype)
    assert p["worlds_per_wavefront"] == K
    assert bw.plan(B, 1, dtype=dtype, one_world=True)["worlds_per_wavefront"] == 1
    assert bw.plan(16 * cus, 1, dtype=dtype)["worlds_per_wavefront"] == 1          # every world has a wavefront anyway
    q, dq, qn, dqn = _tiled(g[kq], g[kdq], B)
    if dtype == torch.float32:
        f = lambda a: np.asarray(a, np.float32).astype(np.float64)
        n0 = len(g[kq]) - 1
        oq, odq, _ = O.step(m, f(q[:n0]), f(dq[:n0]), dt)          # (the batch repeats its first n0 worlds)
        reps = -(-B // n0)
        qn, dqn = np.tile(oq, (reps, 1))[:B], np.tile(odq, (reps, 1))[:B]
    res = {}
    for one in (False, True):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype) if m.nc else None
        bw.step(tq, tdq, dt, 1, cforce=cf, one_world=one)
        torch.cuda.synchronize()
        res[one] = (tq.cpu().numpy(), tdq.cpu().numpy(), None if cf is None else cf.cpu().numpy())
    tol = 1e-8 if dtype == torch.float64 else F32_TOL
    for one in (False, True):
        assert rel(res[one][0], qn) < tol, (one, rel(res[one][0], qn))
        assert rel(res[one][1], dqn) < tol, (one, rel(res[one][1], dqn))
    # forest against one world per wavefront
    ftol = 1e-12 if dtype == torch.float64 else 2e-7
    assert rel(res[False][0], res[True][0]) < ftol and rel(res[False][1], res[True][1]) < ftol
    if m.nc:
        assert rel(res[False][2].reshape(B, -1), res[True][2].reshape(B, -1)) < (1e-9 if dtype == torch.float64 else 1e-5)
    bw.close()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_forest_simplearm_rollout_logs_and_timeline(dtype):
    """simplearm (3 dofs, 8 copies per wavefront): a 64-step launch with a non-uniform timeline and state logs on the
    forest (batch a multiple of the copies) against the one-world launch and, world by world, against the oracle's
    rollout; a batch that is not a multiple logs through the one-world kernels and gives the same states."""
    from arboris_python_amd.batch import BatchedWorlds
    m, q0, dq0 = load_model("simplearm")
    bw = BatchedWorlds(m)
    K = bw.info["forest_copies"]
    assert K == 8
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    B = (16 * cus // K + 7) * K
    rng = np.random.default_rng(4)
    q = np.tile(q0, (B, 1)) + 0.5 * rng.standard_normal((B, m.nq))
    dq = np.tile(dq0, (B, 1)) + 0.5 * rng.standard_normal((B, m.ndof))
    T = 64
    dts = 1e-3 * (1. + 0.5 * np.sin(np.arange(T)))
    out = {}
    for one in (False, True):
        tq, tdq = bw.to_device(q, dq, dtype)
        logs = bw.rollout(tq, tdq, dts, T, log_energy=False, one_world=one)
        torch.cuda.synchronize()
        bw.status()
        out[one] = (tq.cpu().numpy(), tdq.cpu().numpy(), logs["q"].cpu().numpy(), logs["dq"].cpu().numpy())
    ftol = 1e-11 if dtype == torch.float64 else 1e-6
    for a, b in zip(out[False], out[True]):
        assert rel(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])) < ftol
    # the oracle on a sample of worlds (float32: accumulated rounding over 64 steps)
    sel = np.arange(0, B, 97)
    f = (lambda a: a) if dtype == torch.float64 else (lambda a: np.asarray(a, np.float32).astype(np.float64))
    oq, odq, _ = O.rollout(m, f(q[sel]), f(dq[sel]), list(dts))
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    assert rel(out[False][0][sel], oq) < tol and rel(out[False][1][sel], odq) < tol
    # logs hold the state at the BEGINNING of every step
    assert rel(out[False][2][0], f(q)) < 1e-12
    # not a multiple of the copies, with logs: one world per wavefront, same states
    B2 = B - 3
    tq, tdq = bw.to_device(q[:B2], dq[:B2], dtype)
    logs = bw.rollout(tq, tdq, dts, T, log_energy=False)
    torch.cuda.synchronize()
    assert rel(tq.cpu().numpy(), out[True][0][:B2]) < ftol
    assert rel(logs["q"].cpu().numpy()[-1], out[True][2][-1][:B2]) < ftol
    # energies are per world: the launch runs one world per wavefront and still returns them
    tq, tdq = bw.to_device(q, dq, dtype)
    logs = bw.rollout(tq, tdq, 1e-3, 8)
    torch.cuda.synchronize()
    assert logs["energy"].shape == (8, B, 2) and bool(torch.isfinite(logs["energy"]).all())
    bw.close()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, F32_TOL)])
def test_forest_per_world_pd_inputs_and_torques(dtype, tol):
    """One PD controller per world (targets; targets + diagonal gains) and user torques on the forest: every world of the
    golden per-world-PD runs of simplearm, tiled over the wave slots, against the reference."""
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g8_pd_per_world.npz")
    m, _, _ = load_model("simplearm_pdw")
    bw = BatchedWorlds(m)
    K = bw.info["forest_copies"]
    assert K >= 2
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=bw.device).contiguous()
    f = (lambda a: np.asarray(a, np.float64)) if dtype == torch.float64 else (lambda a: np.asarray(a, np.float32).astype(np.float64))
    for tag, with_gains in (("arm_t", False), ("arm_g", True)):
        Q, DQ = g[tag + "_q"], g[tag + "_dq"]
        S, W = Q.shape[0] - 1, Q.shape[1]
        n0 = S * W
        B = 16 * cus + K + 1
        reps = -(-B // n0)
        til = lambda a: np.tile(a, (reps, 1))[:B]
        q, dq = til(Q[:S].reshape(n0, -1)), til(DQ[:S].reshape(n0, -1))
        pd = dict(qdes=til(np.tile(g[tag + "_qdes"], (S, 1))), dqdes=til(np.tile(g[tag + "_dqdes"], (S, 1))))
        if with_gains:
            pd.update(kp=til(np.tile(g[tag + "_kp"], (S, 1))), kd=til(np.tile(g[tag + "_kd"], (S, 1))))
        tau = 0.05 * np.random.default_rng(8).standard_normal((B, m.ndof))
        oq, odq, _ = O.step(m, f(q), f(dq), 5e-3, pd={k: f(v) for k, v in pd.items()}, ext_gforce=f(tau))
        res = {}
        for one in (False, True):
            tq, tdq = bw.to_device(q, dq, dtype)
            bw.step(tq, tdq, 5e-3, 1, pd_targets=(dev(pd["qdes"]), dev(pd["dqdes"])),
                    pd_gains=(dev(pd["kp"]), dev(pd["kd"])) if with_gains else None, ext_gforce=dev(tau), one_world=one)
            torch.cuda.synchronize()
            res[one] = (tq.cpu().numpy(), tdq.cpu().numpy())
            assert rel(res[one][0], oq) < tol and rel(res[one][1], odq) < tol, (tag, one)
        ftol = 1e-12 if dtype == torch.float64 else 2e-7
        assert rel(res[False][0], res[True][0]) < ftol and rel(res[False][1], res[True][1]) < ftol
        # Sick INPUTS (round 4): a NaN / Inf / out-of-range user torque, PD target or gain of one world retires that copy
        # -- NaN state, inputs ignored -- and leaves the worlds that share its wavefront what they are without it.
        rng = np.random.default_rng(21)
        sick = rng.choice(B - K - 1, size=40, replace=False)        # (worlds of the forest launch, not of the remainder)
        tau_s = tau.copy()
        pd_s = {k: v.copy() for k, v in pd.items()}
        for j, w in enumerate(sick):
            kind = j % (4 if with_gains else 3)
            if kind == 0:
                tau_s[w, j % m.ndof] = np.nan
            elif kind == 1:
                pd_s["qdes"][w, j % m.ndof] = np.inf
            elif kind == 2:
                pd_s["dqdes"][w, j % m.ndof] = 3e9 if dtype == torch.float32 else 1e120
            else:
                pd_s["kp"][w, j % m.ndof] = np.nan
        tq, tdq = bw.to_device(q, dq, dtype)
        bw.step(tq, tdq, 5e-3, 1, pd_targets=(dev(pd_s["qdes"]), dev(pd_s["dqdes"])),
                pd_gains=(dev(pd_s["kp"]), dev(pd_s["kd"])) if with_gains else None, ext_gforce=dev(tau_s))
        torch.cuda.synchronize()
        sq, sdq = tq.cpu().numpy(), tdq.cpu().numpy()
        healthy = np.setdiff1d(np.arange(B), sick)
        assert np.isnan(sq[sick]).all() and np.isnan(sdq[sick]).all(), tag
        assert np.isfinite(sq[healthy]).all() and np.isfinite(sdq[healthy]).all(), tag
        assert rel(sq[healthy], res[False][0][healthy]) < ftol and rel(sdq[healthy], res[False][1][healthy]) < ftol, tag
    bw.close()


def test_forest_contact_worlds_through_the_work_queue():
    """A sphere bouncing on a plane (3 translational dofs, one SoftFingerContact): 40 steps in one launch -- the forest
    worlds go through the device-side work queue, forces persist from step to step through cforce -- against the one-world
    launch (bit for bit in float32 here) and the oracle's rollout of a sample."""
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g7_shapes.npz")
    m, _, _ = load_model("shapes_plane_ball")
    bw = BatchedWorlds(m)
    K = bw.info["forest_copies"]
    assert K >= 2
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    B = 16 * cus * K + 5                  # more forest worlds than wave slots: the work queue
    Q, DQ = g["plane_ball_q"], g["plane_ball_dq"]
    rng = np.random.default_rng(11)
    pick = rng.integers(0, len(Q) - 1, size=B)
    q, dq = Q[pick].copy(), DQ[pick].copy()
    dq += 0.05 * rng.standard_normal(dq.shape)
    p = bw.plan(B, 40)
    assert p["worlds_per_wavefront"] == K and p["work_queue"] == 1
    res = {}
    for one in (False, True):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        bw.step(tq, tdq, 5e-3, 40, cforce=cf, one_world=one)
        torch.cuda.synchronize()
        bw.status()
        res[one] = (tq.cpu().numpy(), tdq.cpu().numpy(), cf.cpu().numpy())
    assert float(np.abs(res[True][2]).max()) > 0.1
    # bit for bit the one-world launch (every tree about its own root, segmented prefix sums, and -- round 4 -- the
    # constraint-space products of a copy added in the copy's own groups of four: this model has 3 dofs per copy)
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
    sel = np.arange(0, B, 401)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    oq, odq, _ = O.rollout(m, f(q[sel]), f(dq[sel]), [5e-3] * 40)
    assert rel(res[False][0][sel], oq) < 1e-4 and rel(res[False][1][sel], odq) < 1e-3
    bw.close()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_forest_retires_diverged_copies_without_touching_their_neighbours(dtype):
    """The copies of a forest world share the elimination, where "exact zero x NaN" would spread one copy's NaN to all the
    others.  A copy whose state is not finite (or beyond 1e8 / 1e100) at the beginning of a step is retired instead:
    NaN in its state, forces and logs from then on -- as the one-world kernels leave a world that overflowed -- and
    every other world exactly what it is without the sick ones."""
    from arboris_python_amd.batch import BatchedWorlds
    g = load_golden("g7_shapes.npz")
    m, _, _ = load_model("shapes_plane_ball")
    bw = BatchedWorlds(m)
    K = bw.info["forest_copies"]
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    B = 16 * cus * K                        # (through the work queue: a retired copy stays retired from chunk to chunk)
    Q, DQ = g["plane_ball_q"], g["plane_ball_dq"]
    rng = np.random.default_rng(12)
    pick = rng.integers(0, len(Q) - 1, size=B)
    q, dq = Q[pick].copy(), DQ[pick].copy()
    sick = rng.choice(B, size=200, replace=False)
    kinds = rng.integers(0, 4, size=len(sick))
    qs, dqs = q.copy(), dq.copy()
    for w, kind in zip(sick, kinds):
        if kind == 0:
            qs[w, 3] = np.nan                # a NaN position
        elif kind == 1:
            dqs[w, 4] = np.inf               # an infinite velocity
        elif kind == 2:
            dqs[w, 1] = 3e9 if dtype == torch.float32 else 1e120   # diverged: beyond the retirement range
        else:
            qs[w, :] = np.nan
    T = 12
    healthy = np.setdiff1d(np.arange(B), sick)
    out = {}
    for name, (qq, dd) in (("clean", (q, dq)), ("sick", (qs, dqs))):
        tq, tdq = bw.to_device(qq, dd, dtype)
        cf = bw.new_cforce(B, dtype)
        assert bw.plan(B, T, dtype=dtype)["worlds_per_wavefront"] == K
        logs = bw.rollout(tq, tdq, 5e-3, T, cforce=cf, log_energy=False)
        torch.cuda.synchronize()
        bw.status()
        out[name] = [t.cpu().numpy() for t in (tq, tdq, cf, logs["q"], logs["dq"])]
    # (to rounding, not bit for bit: a retired copy's state of rest changes the prefix sums the composites of the copies
    # after it are differences of, in float64)
    tol = 1e-11 if dtype == torch.float64 else 1e-5
    for k, (a, b) in enumerate(zip(out["clean"], out["sick"])):
        if a.ndim == 3 and a.shape[0] == T:          # logs [step][world][..]
            a, b = a[:, healthy].reshape(T * len(healthy), -1), b[:, healthy].reshape(T * len(healthy), -1)
        else:
            a, b = a[healthy].reshape(len(healthy), -1), b[healthy].reshape(len(healthy), -1)
        assert np.isfinite(b).all()
        assert rel(b, a) < (tol if k != 2 else 1e3 * tol), (k, rel(b, a))
    sq, sdq, scf, lq, ldq = out["sick"]
    assert np.isnan(sq[sick]).all() and np.isnan(sdq[sick]).all() and np.isnan(scf[sick]).all()
    assert np.isnan(lq[1:, sick]).all() and np.isnan(ldq[1:, sick]).all()
    assert np.isfinite(sq[healthy]).all()
    # a world that diverges DURING the launch: huge but finite velocity, retired at the next step it exceeds the range
    q2, dq2 = q.copy(), dq.copy()
    w = int(healthy[7])
    dq2[w, 3:] = 2e7 if dtype == torch.float32 else 1e60
    tq, tdq = bw.to_device(q2, dq2, dtype)
    cf = bw.new_cforce(B, dtype)
    bw.step(tq, tdq, 5e-3, T, cforce=cf)
    torch.cuda.synchronize()
    r = tq.cpu().numpy()
    others = np.setdiff1d(np.arange(B), [w])
    assert np.isfinite(r[others]).all() and rel(r[others], out["clean"][0][others]) < tol
    bw.close()
