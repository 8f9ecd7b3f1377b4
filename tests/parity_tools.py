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
      and that system is the oracle's to float32 accuracy: Y' within SYS_TOL (round 4: 1e-5 of its largest entry, twice the
      largest value measured on the outliers that reach this criterion), each of the flipped solve's own rows of Y' within ROW_TOL of the oracle's row (relative to that row's
      largest entry), v' within VEL_TOL (2.5e-5, likewise twice the measured maximum).  A bug in the sweeps fails the first
      half, a bug in the dynamics or the elimination the second.  Or
  (e) float64 sweeps on the device's system take the ORACLE's decision at that solve, but the decision lies inside the
      running rounding-error bound of ANY float32 execution of the sweeps: every update v' += Y'[:, c] df (core.py:935)
      commits at most one float32 ulp of its terms' magnitudes per row (the terms are kN forces times 1e-3 admittances
      that cancel to the small velocity the solve then tests), and with the velocity rows of that solve moved within the
      accumulated bound the oracle's own solve decides like the device (`solve_samples` random patterns).  An error of
      the sweeps larger than float32 rounding still fails.

Anything else is reported as unexplained (None) and the callers fail.

Every reason is a `Reason` (a str with `.criterion`: "active", "a", "b", "c", "d", "e", "ill"), `check_replay` counts
the outliers per criterion and caps the share of world-steps that (d) and (e) -- the criteria that argue from the
DEVICE's own system -- may explain (DE_SHARE_CAP).
"""
import numpy as np

import arb_oracle as O

MARGIN_TOL = 1e-5
# Tolerances of criterion (d) / (e): "the device's system is the oracle's to float32 accuracy".  Measured over 159 744
# world-steps (profiles/r04_replay_stats.txt), on the outliers that reach the criterion -- the tail of the distribution:
# Y' up to 5.2e-6 of its largest entry, the flipped solve's own rows up to 5.2e-6 of theirs, v' up to 1.3e-5 of its
# largest entry (v' = J' x the free velocity, whose own gate is 1e-5, through rows of J' whose absolute sums are 3-5).
# The tolerances sit above those maxima (Y' and the rows 2 x, v' 1.4 x: 1.8e-5 on one world-step of the sample); round 3 had 2e-5 on Y' alone, which a dynamics error of 1e-5 of Y' would have
# passed; 2e-6 (the first round-4 value) left 1-3 world-steps per 40 000 unexplained.
SYS_TOL = 1e-5        # |Y'_device - Y'_oracle| / max|Y'_oracle|            (rounds 3-4: fitted to the measured tail.  Since round 6
ROW_TOL = 1e-5        # the flipped solve's own four rows of Y', each relative to its own largest entry       DIAGNOSTICS only:
VEL_TOL = 2.5e-5      # |v'_device - v'_oracle| / max|v'_oracle|                                 the a-priori bounds below decide)
# Round 5: the tolerances of criterion (d) / (e) come from an A-PRIORI bound, not from the device's measured errors.  The device
# forms Y' = J' Y J'^T and v' = J' Y r in float32 (unit round-off u = 2^-24) through a pivot-free elimination of the n x n
# impedance matrix.  First-order backward-error analysis (Higham, Accuracy and Stability of Numerical Algorithms, Thm 9.3 for the
# elimination, Lemma 3.5 for the products): every computed entry differs from the exact one by at most
#       gamma (|J'| |Y| |J'|^T)_ij      resp.      gamma (|J'| (|gvel| + |Y| |r|))_i,          gamma = APRIORI_K n u,
# (r = gforce - (N + B) gvel: the device solves the increment form of core.py:975-976)
# with the entrywise absolute values of the ORACLE's own float64 factors -- a bound that knows nothing about the device's
# errors.  APRIORI_K = 4: three roundings per term of the triple product plus the elimination's 3 n u growth-free bound,
# rounded up.  A device system inside these bounds is "the oracle's system to float32 accuracy"; outside, criterion (d) / (e)
# do not apply whatever the fitted tolerances say.  (human36: gamma = 4 x 42 x 2^-24 = 1.0e-5 of the bounding products.)
APRIORI_K = 4.
U32 = 2. ** -24
DE_SHARE_CAP = 5e-4   # criteria (d) + (e) may explain at most 0.05 % of the replayed world-steps (measured: 4 contacts 0.008 -
                      # 0.025 %; 8 contacts on body-space columns, the default since round 5, 0 - 0.005 % -- until round 5 the callers
                      # passed 4e-3 for 8 contacts, what TWO column sets had measured: 0.17 - 0.20 %)
F32_TOL = 1e-5


LAST_DIAG = {}        # what the last explain_outlier call measured (printed by check_replay when no criterion holds)
KERNEL_KW = {}        # kernel-selection keywords of BatchedWorlds.step / inspect (body_columns, general_kernels) of the rollout
                      # under adjudication: the inspect kernel must form the constraint-space system as the step kernel did


class Reason(str):
    """An explanation with the criterion that produced it."""
    def __new__(cls, criterion, text):
        r = super().__new__(cls, text)
        r.criterion = criterion
        return r


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
    r = bw.inspect(tq, tdq, dt, ["gs_stats", "gs_trace", "c_active", "c_adm", "c_vel", "c_sdist", "c_jac"], cforce=bw.new_cforce(1, torch.float32),
                   **KERNEL_KW)
    st = r["gs_stats"].cpu().numpy()[0]                    # release, static, fast slide, eig6 slide, sweeps
    dtr = r["gs_trace"].cpu().numpy()[0]                   # (20, nc): decision of every executed solve, -1 = not run
    dact = r["c_active"].cpu().numpy()[0].astype(bool)
    tr, d = _trace_of(m, q, dq, dt)
    rng = np.random.default_rng(seed)
    LAST_DIAG.clear()
    if not np.array_equal(dact, d["active"][0]):
        # the activity test sd + dsd dt < proximity (constraints.py:292) went the other way for some contact
        diff = np.flatnonzero(dact != d["active"][0])
        pred, prox = d["gap_pred"][0][diff], np.asarray(m.c_prox)[diff]
        mg = np.abs(pred - prox) / np.maximum(np.abs(d["sdist"][0][diff]) + np.abs(pred - d["sdist"][0][diff]) + prox, 1e-300)
        if mg.max() < MARGIN_TOL:
            return Reason("active", "active set differs (contact %s), the oracle's activity test within %.1e of its threshold" % (diff.tolist(), mg.max()))
        for _ in range(samples):
            pq, pdq = _perturbed(q, dq, rng)
            _, dp = _trace_of(m, pq, pdq, dt)
            if not np.array_equal(dp["active"][0], d["active"][0]):
                return Reason("active", "active set differs (contact %s), and the oracle's own active set changes under a one-ulp input change" % diff.tolist())
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
        return Reason("a", "decision differs at %s: the oracle's inequality within %.1e of equality" % (where, mg))
    key = lambda x: (x["sweep"], x["c"])
    base = [(key(x), x["branch"]) for x in tr[:i + 1]]
    for _ in range(samples):
        pq, pdq = _perturbed(q, dq, rng)
        trp, _ = _trace_of(m, pq, pdq, dt)
        if [(key(x), x["branch"]) for x in trp[:i + 1]] != base:
            return Reason("b", "decision differs at %s (margin %.1e): the oracle's own trace changes at or before that solve "
                               "under a one-ulp input change" % (where, mg))
    if np.all(np.asarray(m.c_eps)[t["c"]] == 1.) and _solve_flips(t, rng, solve_samples):
        return Reason("c", "decision differs at %s (margin %.1e): the oracle's own decision at that solve changes under one-ulp "
                           "(float32) changes of the solve's inputs" % (where, mg))
    if np.all(np.asarray(m.ctype) == 0):
        # (d) float64 sweeps on the device's own float32 system
        adm_d = r["c_adm"].double().cpu().numpy()[0]; vel_d = r["c_vel"].double().cpu().numpy()[0]
        sd_d = r["c_sdist"].double().cpu().numpy()[0]
        e_adm = np.abs(adm_d - d["adm"][0]).max() / max(np.abs(d["adm"][0]).max(), 1e-300)
        e_vel = 0.
        if d.get("vel0") is not None:       # v' the sweeps start from (core.py:925)
            e_vel = float(np.abs(vel_d - d["vel0"][0]).max() / max(np.abs(d["vel0"][0]).max(), 1e-300))
        rows = slice(4 * t["c"], 4 * t["c"] + 4)
        e_row = float((np.abs(adm_d[rows] - d["adm"][0][rows]).max(axis=1) / np.maximum(np.abs(d["adm"][0][rows]).max(axis=1), 1e-300)).max())
        # the a-priori float32 bounds of Y' and v' (see APRIORI_K): entrywise, from the oracle's Y = Z^-1 and rhs and the constraint
        # Jacobian (the device's rows of J', float32 roundings of the oracle's: good enough for |J'|)
        gam = APRIORI_K * m.ndof * U32
        Jc = np.abs(r["c_jac"].double().cpu().numpy()[0].reshape(-1, m.ndof))
        Yabs = np.abs(d["Y"][0])
        B_adm = gam * (Jc @ Yabs @ Jc.T)
        # (v': the device solves the INCREMENT form Z (gvel+ - gvel) = gforce - (N + B) gvel and adds gvel back -- DESIGN.md 2 --,
        # so the terms that go through the elimination are those of the increment, not M gvel / dt)
        dq64 = np.asarray(dq, np.float64)
        rhs_inc = d["gforce0"][0] - (d["N"][0] + d["Bv"][0]) @ dq64
        B_vel = gam * (Jc @ (np.abs(dq64) + Yabs @ np.abs(rhs_inc)))
        act_rows = np.repeat(dact, 4)
        in_adm = bool(np.all(np.abs(adm_d - d["adm"][0])[np.ix_(act_rows, act_rows)] <= B_adm[np.ix_(act_rows, act_rows)] + 1e-300))
        in_vel = d.get("vel0") is None or bool(np.all(np.abs(vel_d - d["vel0"][0])[act_rows] <= B_vel[act_rows] + 1e-300))
        # (round 6: the A-PRIORI bounds ALONE decide "the oracle's system to float32 accuracy"; the tolerances fitted to the
        # measured tail in rounds 3-4 -- SYS_TOL / ROW_TOL / VEL_TOL, 1.4-2 x one sample's maxima -- were flake-prone caps on top
        # of a bound that needs no fitting, and are kept as diagnostics only)
        sys_ok = in_adm and in_vel
        LAST_DIAG.update(where=where, margin=float(mg), e_adm=float(e_adm), e_vel=e_vel, e_row=e_row, in_apriori_bound=bool(in_adm and in_vel),
                         apriori_adm=float(B_adm[np.ix_(act_rows, act_rows)].max() / max(np.abs(d["adm"][0]).max(), 1e-300)),
                         apriori_vel=float(B_vel[act_rows].max() / max(np.abs(d["vel0"][0]).max(), 1e-300)) if d.get("vel0") is not None else 0.)
        got = sweeps_on(m, adm_d, vel_d, sd_d, dact, dt, nsweeps=t["sweep"] + 1)
        same = all(min(int(dtr[s_, c_]), 2) == got[s_, c_] for s_ in range(t["sweep"] + 1) for c_ in range(m.nc)
                   if dact[c_] and (s_ < t["sweep"] or c_ <= t["c"]))
        LAST_DIAG.update(device_trace_reproduced=bool(same))
        if same and sys_ok:
            return Reason("d", "decision differs at %s (margin %.1e): float64 sweeps on the device's own float32 system take the device's "
                               "decisions; that system is within %.1e (Y'), %.1e (v') of the oracle's (the solve's own rows %.1e)" % (where, mg, e_adm, e_vel, e_row))
        # (e) the decision lies inside the rounding-error bound of float32 sweeps on that system
        before_ok = all(min(int(dtr[s_, c_]), 2) == got[s_, c_] for s_ in range(t["sweep"] + 1) for c_ in range(m.nc)
                        if dact[c_] and (s_ < t["sweep"] or c_ < t["c"]))
        if before_ok and sys_ok:
            _, at = sweeps_on(m, adm_d, vel_d, sd_d, dact, dt, nsweeps=t["sweep"] + 1, stop_at=(t["sweep"], t["c"]))
            if at is not None:
                ulp = lambda a: np.asarray(a, np.float64) * (1. + 2. ** -24 * rng.choice([-1., 1.], np.shape(a)))
                for _ in range(solve_samples):
                    v = at["vel"] + at["bound"] * rng.uniform(-1., 1., 4)
                    br = O._softfinger_solve_one(v, ulp(at["adm"]), ulp(at["force"]), at["sdist"], at["mu"], np.ones(3), dt)[2]
                    if br == dev:
                        return Reason("e", "decision differs at %s (margin %.1e): inside the running rounding-error bound of float32 sweeps "
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
        return Reason("ill", "ill-conditioned step: one float32 ulp on the input moves the oracle by q %.1e dq %.1e" % (sq, sdq))
    return None


def replay_errors(m, log_q, log_dq, steps, worlds, dt, with_index=False, ext=None):
    """Oracle step from the device's own logged state at `steps` for `worlds` (`ext`: the worlds' user torques, (B, ndof));
    returns the per-world errors of q and dq against the device's next logged state, stacked over the steps (and, with
    `with_index`, the (step, world) pair of every entry)."""
    eq, edq, idx = [], [], []
    for k in steps:
        q = log_q[k][worlds].double().cpu().numpy()
        dq = log_dq[k][worlds].double().cpu().numpy()
        kw = {} if ext is None else dict(ext_gforce=np.asarray(ext, np.float64)[worlds])
        oq, odq, _ = O.step(m, q, dq, dt, **kw)
        eq.append(world_err(log_q[k + 1][worlds].cpu().numpy(), oq))
        edq.append(world_err(log_dq[k + 1][worlds].cpu().numpy(), odq))
        idx += [(k, int(w)) for w in worlds]
    if with_index:
        return np.concatenate(eq), np.concatenate(edq), idx
    return np.concatenate(eq), np.concatenate(edq)


def check_replay(bw, m, log, steps, worlds, dt, min_ok=0.995, max_outlier=1e-3, verbose=True, ext=None, de_cap=None):
    """Replay `steps` x `worlds` of a logged float32 rollout through the float64 oracle and adjudicate EVERY world-step over
    the gate.  Fails on: fewer than `min_ok` of the world-steps within F32_TOL; an outlier no criterion explains; an
    outlier with identical decisions above `max_outlier` (q) / 10 x that (dq); more than DE_SHARE_CAP of the world-steps
    explained by criteria (d) / (e).  A world-step whose decisions differ from the oracle's (criteria active, a-e) may
    exceed `max_outlier` -- a contact that releases instead of sliding changes dq+ by whatever its force was worth -- and
    is printed with its reason.  Returns a dict: ok share, largest errors, outliers per criterion, cases above the cap."""
    if ext is not None:
        raise NotImplementedError("adjudication of outliers with user torques: replay_errors(ext=) gives the errors")
    eq, edq, idx = replay_errors(m, log["q"], log["dq"], steps, worlds, dt, with_index=True)
    ok = (eq < F32_TOL) & (edq < F32_TOL)
    assert ok.mean() >= min_ok, (ok.mean(), eq.max(), edq.max())
    counts, over = {}, []
    for i in np.flatnonzero(~ok):
        k, w = idx[i]
        qk, dqk = log["q"][k][w].cpu().numpy(), log["dq"][k][w].cpu().numpy()
        why = explain_outlier(bw, m, qk, dqk, dt)
        if why is None:
            # same decisions everywhere: only acceptable when the step itself is that ill-conditioned -- the float64 oracle
            # moves by at least the observed error under a one-ulp (float32) input change
            why = ill_conditioned(m, qk, dqk, dt, eq[i], edq[i], cap=max_outlier)
        assert why is not None, "unexplained outlier: step %d world %d, err q %.2e dq %.2e; %s" % (k, w, eq[i], edq[i], LAST_DIAG)
        counts[why.criterion] = counts.get(why.criterion, 0) + 1
        big = not (eq[i] < max_outlier and edq[i] < 10 * max_outlier)
        if big:
            assert why.criterion != "ill", (k, w, eq[i], edq[i], why)
            over.append((k, w, float(eq[i]), float(edq[i]), str(why)))
        if verbose or big:
            print("outlier step %d world %d: err q %.2e dq %.2e [%s]%s -- %s"
                  % (k, w, eq[i], edq[i], why.criterion, " ABOVE THE CAP" if big else "", why))
    n = len(ok)
    de = counts.get("d", 0) + counts.get("e", 0)
    de_cap = DE_SHARE_CAP if de_cap is None else de_cap
    assert de <= max(1, int(np.ceil(de_cap * n))), ("criteria (d)/(e) explain %d of %d world-steps" % (de, n), counts)
    return dict(ok=float(ok.mean()), n=n, max_q=float(eq.max()), max_dq=float(edq.max()), criteria=counts, over_cap=over)
