"""Shared machinery of the float32 parity gates (tests/test_gpu_full_size.py, tools/replay_stats.py).

A float32 device step may differ from the float64 oracle by more than rounding ONLY when the step contains a
decision (the contacts' activity test, constraints.py:292; the release test and the friction-cone test of
SoftFingerContact.solve, constraints.py:781, 799) that the device and the oracle took differently AND that decision
is MARGINAL FOR THE ORACLE ITSELF.  "The traces differ" is not an explanation by itself: a kernel bug that flips a
comfortable decision must fail.  `explain_outlier` therefore demands, at the FIRST solve where the device's trace
(arb_inspect_out.gs_trace) leaves the oracle's,

  (a) the oracle's own inequality at that solve sits within MARGIN_TOL (1e-5: the parity tolerance) of equality,
      measured against the magnitude of the terms the inequality is computed from (`solve_margins`), or
  (b) the oracle's own trace changes at or before that solve when its INPUT state moves by one float32 ulp
      (`samples` random sign patterns, conftest.oracle_sensitivity's perturbation).

Anything else is reported as unexplained (None) and the callers fail.
"""
import numpy as np

import arb_oracle as O

MARGIN_TOL = 1e-5


def world_err(a, b):
    """Per-world relative error (B,)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b), axis=1) / np.maximum(1., np.max(np.abs(b), axis=1))


def solve_margins(t):
    """Relative distances of the two inequalities of ONE SoftFingerContact.solve call (an entry of the oracle's
    trace) from equality: (release test constraints.py:781, friction-cone test :799 or None when the solve released).
    Each distance is divided by the first-order bound of what a unit RELATIVE perturbation of the solve's inputs
    (vel, admittance, force, sdist) can move the tested quantity by -- the sum of the magnitudes of the terms it is
    computed from -- so "< 1e-5" reads: the decision flips under an input perturbation of the size of the parity
    tolerance."""
    vel, adm, f, sd, dt, mu = t["vel"], np.asarray(t["adm"]), t["force"], t["sdist"], t["dt"], t["mu"]
    v0n = vel[3] - adm[3] @ f
    rel = sd + dt * v0n
    rel_scale = abs(sd) + dt * (abs(vel[3]) + np.abs(adm[3] * f).sum())
    m_rel = abs(rel) / max(rel_scale, 1e-300)
    if rel > 0:
        return m_rel, None
    P = np.linalg.pinv(adm)
    tgt = np.hstack((vel[0:3], vel[3] + sd / dt))
    fn = f - P @ tgt
    fn_scale = np.abs(f) + np.abs(P) @ np.abs(np.hstack((vel[0:3], abs(vel[3]) + abs(sd / dt))))
    lhs, rhs = float(np.sum(fn[0:3] ** 2)), float((fn[3] * mu) ** 2)
    cone_scale = 2. * float(np.sum(np.abs(fn[0:3]) * fn_scale[0:3])) + 2. * mu * mu * abs(fn[3]) * fn_scale[3]
    return m_rel, abs(lhs - rhs) / max(cone_scale, 1e-300)


def _trace_of(m, q, dq, dt):
    tr = []
    _, _, _, d = O.step(m, q[None].astype(np.float64), dq[None].astype(np.float64), dt, debug=True, trace=tr)
    return tr, d


def _perturbed(q, dq, rng):
    pq = q.astype(np.float64) * (1. + 2. ** -24 * rng.choice([-1., 1.], q.shape))
    pdq = dq.astype(np.float64) * (1. + 2. ** -24 * rng.choice([-1., 1.], dq.shape))
    return pq, pdq


def explain_outlier(bw, m, q, dq, dt, samples=8, seed=0):
    """Why may the float32 step from (q, dq) (one world, the float32 values the device stepped from) differ from the
    float64 oracle by more than rounding?  Returns a reason string that PROVES a marginal decision, or None."""
    import torch
    tq = torch.as_tensor(q[None], dtype=torch.float32, device=bw.device).contiguous()
    tdq = torch.as_tensor(dq[None], dtype=torch.float32, device=bw.device).contiguous()
    r = bw.inspect(tq, tdq, dt, ["gs_stats", "gs_trace", "c_active"], cforce=bw.new_cforce(1, torch.float32))
    st = r["gs_stats"].cpu().numpy()[0]                    # release, static, fast slide, eig6 slide, sweeps
    dtr = r["gs_trace"].cpu().numpy()[0]                   # (20, nc): decision of every executed solve, -1 = not run
    dact = r["c_active"].cpu().numpy()[0].astype(bool)
    tr, d = _trace_of(m, q, dq, dt)
    rng = np.random.default_rng(seed)
    if not np.array_equal(dact, d["active"][0]):
        # the activity test sd + dsd dt < proximity (constraints.py:292) went the other way for some contact
        diff = np.flatnonzero(dact != d["active"][0])
        pred, prox = d["gap_pred"][0][diff], np.asarray(m.c_prox)[diff]
        mg = np.abs(pred - prox) / np.maximum(np.abs(d["sdist"][0][diff]) + np.abs(pred - d["sdist"][0][diff]) + prox, 1e-300)
        if mg.max() < MARGIN_TOL:
            return "active set differs (contact %s), the oracle's activity test within %.1e of its threshold" % (diff.tolist(), mg.max())
        for _ in range(samples):
            pq, pdq = _perturbed(q, dq, rng)
            _, dp = _trace_of(m, pq, pdq, dt)
            if not np.array_equal(dp["active"][0], d["active"][0]):
                return "active set differs (contact %s), and the oracle's own active set changes under a one-ulp input change" % diff.tolist()
        return None
    # the device stops sweeping at a bit-exact fixed point: compare the sweeps it executed, solve by solve
    nsw = int(st[4])
    first = None
    for i, t in enumerate(tr):
        if t["sweep"] < nsw:
            dev = int(dtr[t["sweep"], t["c"]])
            if min(dev, 2) != t["branch"]:
                first = (i, t, min(dev, 2))
                break
    if first is None:
        return None
    i, t, dev = first
    m_rel, m_cone = solve_margins(t)
    # the inequality that separates the two outcomes: the release test when either side released, else the cone test
    mg = m_rel if (t["branch"] == 0 or dev == 0 or m_cone is None) else m_cone
    where = "sweep %d contact %d (oracle %d, device %d)" % (t["sweep"], t["c"], t["branch"], dev)
    if mg < MARGIN_TOL:
        return "decision differs at %s: the oracle's inequality within %.1e of equality" % (where, mg)
    key = lambda x: (x["sweep"], x["c"])
    base = [(key(x), x["branch"]) for x in tr[:i + 1]]
    for _ in range(samples):
        pq, pdq = _perturbed(q, dq, rng)
        trp, _ = _trace_of(m, pq, pdq, dt)
        if [(key(x), x["branch"]) for x in trp[:i + 1]] != base:
            return ("decision differs at %s (margin %.1e): the oracle's own trace changes at or before that solve "
                    "under a one-ulp input change" % (where, mg))
    return None
