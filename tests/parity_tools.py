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
      (`samples` random sign patterns, conftest.oracle_sensitivity's perturbation), or
  (c) the oracle's own decision AT THAT SOLVE changes when the solve's inputs (vel, admittance block, force, sdist: the
      arguments of SoftFingerContact.solve, constraints.py:780) move by one float32 ulp each -- the least any float32
      implementation perturbs them, since it stores them in float32 (`solve_samples` random sign patterns), or
  (d) the sweeps are RIGHT and the float32 SYSTEM moved the decision: the oracle's own solve (constraints.py:780-836) run
      in float64, sweep by sweep in the reference's order, on the constraint-space system the DEVICE built (Y' = J' Y J'^T
      and v', arb_inspect_out.c_adm / c_vel, float32) takes the device's decisions up to and including that solve --
      and that system is within SYS_TOL of the oracle's (relative to its largest entry).  A bug in the sweeps fails the
      first half, a bug in the dynamics or the elimination the second.  Or
  (e) float64 sweeps on the device's system take the ORACLE's decision at that solve, but the decision lies inside the
      running rounding-error bound of ANY float32 execution of the sweeps: every update v' += Y'[:, c] df (core.py:935)
      commits at most one float32 ulp of its terms' magnitudes per row (the terms are kN forces times 1e-3 admittances
      that cancel to the small velocity the solve then tests), and with the velocity rows of that solve moved within the
      accumulated bound the oracle's own solve decides like the device (`solve_samples` random patterns).  An error of
      the sweeps larger than float32 rounding still fails.

Anything else is reported as unexplained (None) and the callers fail.
"""
import numpy as np

import arb_oracle as O

MARGIN_TOL = 1e-5
SYS_TOL = 2e-5        # |Y'_device - Y'_oracle| / max|Y'_oracle| and the same for v' (float32 elimination at cond(Z) ~ 7e4)


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


def _solve_flips(t, rng, samples):
    """Does the oracle's decision at the traced solve `t` change under one-ulp (float32) changes of the solve's inputs?"""
    eps = np.ones(3)
    ulp = lambda a: np.asarray(a, np.float64) * (1. + 2. ** -24 * rng.choice([-1., 1.], np.shape(a)))
    for _ in range(samples):
        br = O._softfinger_solve_one(ulp(t["vel"]), ulp(t["adm"]), ulp(t["force"]), float(ulp(t["sdist"])), t["mu"], eps, t["dt"])[2]
        if br != t["branch"]:
            return True
    return False


def sweeps_on(m, adm, vel, sdist, active, dt, nsweeps=20, stop_at=None):
    """World.update_constraints' Gauss-Seidel loop (core.py:929-935) with SoftFingerContact.solve (constraints.py:780-836)
    in float64 on a GIVEN constraint-space system: adm (R, R), vel (R,), sdist / active (nc,).  Returns the decision of
    every solve as an (nsweeps, nc) array (-1: constraint inactive).  With `stop_at = (sweep, c)` it stops BEFORE that
    solve and also returns its inputs and the running float32 rounding-error bound of the velocity rows: every update
    v' += Y'[:, c] df done in float32 commits at most 2^-24 (|v'| + sum_i |Y'[:, c_i] df_i|) per row, accumulated."""
    nc = m.nc
    vel = np.array(vel, np.float64); adm = np.asarray(adm, np.float64)
    force = np.zeros((nc, 4))
    bound = 2. ** -24 * np.abs(vel)                    # v' itself is a float32 number
    out = -np.ones((nsweeps, nc), int)
    for k in range(nsweeps):
        for c in range(nc):
            if not active[c]:
                continue
            rows = slice(4 * c, 4 * c + 4)
            if stop_at is not None and (k, c) == tuple(stop_at):
                return out, dict(vel=vel[rows].copy(), adm=adm[rows, rows].copy(), force=force[c].copy(), sdist=float(sdist[c]),
                                 mu=float(m.c_mu[c]), dt=dt, bound=bound[rows].copy())
            df, newf, br = O._softfinger_solve_one(vel[rows], adm[rows, rows], force[c], float(sdist[c]), float(m.c_mu[c]),
                                                   np.asarray(m.c_eps[c], float), dt)
            out[k, c] = br
            force[c] = newf
            inc = adm[:, rows] * df[None, :]
            vel += inc.sum(axis=1)
            bound += 2. ** -24 * (np.abs(inc).sum(axis=1) + np.abs(vel))
    return out if stop_at is None else (out, None)


def explain_outlier(bw, m, q, dq, dt, samples=8, seed=0, solve_samples=64):
    """Why may the float32 step from (q, dq) (one world, the float32 values the device stepped from) differ from the
    float64 oracle by more than rounding?  Returns a reason string that PROVES a marginal decision, or None."""
    import torch
    tq = torch.as_tensor(q[None], dtype=torch.float32, device=bw.device).contiguous()
    tdq = torch.as_tensor(dq[None], dtype=torch.float32, device=bw.device).contiguous()
    r = bw.inspect(tq, tdq, dt, ["gs_stats", "gs_trace", "c_active", "c_adm", "c_vel", "c_sdist"], cforce=bw.new_cforce(1, torch.float32))
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
    if np.all(np.asarray(m.c_eps)[t["c"]] == 1.) and _solve_flips(t, rng, solve_samples):
        return ("decision differs at %s (margin %.1e): the oracle's own decision at that solve changes under one-ulp "
                "(float32) changes of the solve's inputs" % (where, mg))
    if np.all(np.asarray(m.ctype) == 0):
        # (d) float64 sweeps on the device's own float32 system
        adm_d = r["c_adm"].double().cpu().numpy()[0]; vel_d = r["c_vel"].double().cpu().numpy()[0]
        sd_d = r["c_sdist"].double().cpu().numpy()[0]
        e_adm = np.abs(adm_d - d["adm"][0]).max() / max(np.abs(d["adm"][0]).max(), 1e-300)
        got = sweeps_on(m, adm_d, vel_d, sd_d, dact, dt, nsweeps=t["sweep"] + 1)
        same = all(min(int(dtr[s_, c_]), 2) == got[s_, c_] for s_ in range(t["sweep"] + 1) for c_ in range(m.nc)
                   if dact[c_] and (s_ < t["sweep"] or c_ <= t["c"]))
        if same and e_adm < SYS_TOL:
            return ("decision differs at %s (margin %.1e): float64 sweeps on the device's own float32 system take the device's "
                    "decisions; that system is within %.1e of the oracle's" % (where, mg, e_adm))
        # (e) the decision lies inside the rounding-error bound of float32 sweeps on that system
        before_ok = all(min(int(dtr[s_, c_]), 2) == got[s_, c_] for s_ in range(t["sweep"] + 1) for c_ in range(m.nc)
                        if dact[c_] and (s_ < t["sweep"] or c_ < t["c"]))
        if before_ok and e_adm < SYS_TOL:
            _, at = sweeps_on(m, adm_d, vel_d, sd_d, dact, dt, nsweeps=t["sweep"] + 1, stop_at=(t["sweep"], t["c"]))
            if at is not None:
                ulp = lambda a: np.asarray(a, np.float64) * (1. + 2. ** -24 * rng.choice([-1., 1.], np.shape(a)))
                for _ in range(solve_samples):
                    v = at["vel"] + at["bound"] * rng.uniform(-1., 1., 4)
                    br = O._softfinger_solve_one(v, ulp(at["adm"]), ulp(at["force"]), at["sdist"], at["mu"], np.ones(3), dt)[2]
                    if br == dev:
                        return ("decision differs at %s (margin %.1e): inside the running rounding-error bound of float32 sweeps "
                                "(velocity rows of that solve known to +-%.1e of %.1e)" % (where, mg, at["bound"].max(), np.abs(at["vel"]).max()))
    return None


def ill_conditioned(m, q, dq, dt, eq, edq, cap=1e-3, samples=8):
    """Identical decisions, yet an error above the gate: accepted only when the step is that ill-conditioned FOR THE
    ORACLE -- its own (q+, dq+) move by at least the observed error when its input moves by one float32 ulp (the
    worst of `samples` sign patterns) -- and the error stays below `cap`.  Returns a reason string or None."""
    from conftest import oracle_sensitivity
    sq, sdq = oracle_sensitivity(m, q[None], dq[None], dt, samples=samples)
    sq, sdq = float(sq[0]), float(sdq[0])
    if sq >= eq and sdq >= edq and max(eq, edq) < cap:
        return "ill-conditioned step: one float32 ulp on the input moves the oracle by q %.1e dq %.1e" % (sq, sdq)
    return None
