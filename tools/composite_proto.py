#!/usr/bin/env python3
"""NumPy prototype of the kernel's composite phase B, formula by formula as the HIP code
computes them (3x3 blocks, world axes about the root body's position), checked against the
oracle: Z, the increment right-hand side and the constraint Jacobian rows.  CPU only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import arb_oracle as O
from conftest import load_model
from arboris_python_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "human36_c4"
m, q0, dq0 = load_model(name)
if name.startswith("shapes") or name in ("ballsocket",):
    q, dq = q0[None].copy(), dq0[None].copy()
    dq = dq + 0.3
else:
    q, dq = synth.random_states(m, 2, seed=5)
dt = 5e-3
dyn = O.update_dynamic(m, q, dq)
gforce, Zfull, Y = O.update_controllers(m, dyn, q, dq, dt)
Zref = dyn["M"] / dt + dyn["Bv"] + dyn["N"]
rhs_ref = gforce - ((dyn["N"] + dyn["Bv"]) @ dq[..., None])[..., 0]
n, nb = m.ndof, m.nb
hat = O.hat
cross = np.cross
for w in range(q.shape[0]):
    body_of = np.zeros(n, int)
    for b in range(nb):
        body_of[m.dof_off[b]:m.dof_off[b] + m.jnd[b]] = b
    # ---- phase A products (body frame): twist, pseudo twist Om, wrench PT --------------------
    Om = np.zeros((nb, 6)); PT = np.zeros((nb, 6))
    g6 = np.zeros(6); g6[3:6] = m.gravity
    for b in range(nb):
        p = int(m.parent[b])
        ds = slice(int(m.dof_off[b]), int(m.dof_off[b] + m.jnd[b])); qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
        jt = int(m.jtype[b])
        H_rn, J_nr, dJ_nr = O.joint_kinematics(jt, q[w:w + 1, qs], dq[w:w + 1, ds])
        T_nr = dq[w, ds].copy() if jt == O.JT_FREE else (J_nr[0] @ dq[w, ds])
        Ad_nr = O.adjoint(O.joint_ipose(jt, q[w:w + 1, qs], H_rn))[0]
        H_pc = m.H_pr[b] @ (H_rn[0] @ O.hinv(m.H_cn[b]))
        W = O.iadjoint(H_pc) @ (O.adjoint(m.H_pr[b]) @ (-(Ad_nr @ T_nr)))      # = Ad_cn Ad_nr T_rn
        Om[b] = W + (0 if p < 0 else O.iadjoint(H_pc) @ Om[p])
        tw = dyn["twist"][w, b]
        ab = dyn["djac"][w, b] @ dq[w]
        grav = O.iadjoint(dyn["pose"][w, b]) @ g6 if m.weighted[b] else np.zeros(6)
        PT[b] = m.mass[b] @ grav - m.mass[b] @ ab - dyn["nle"][w, b] @ tw - m.visc[b] @ tw
    # ---- B2.1 lane = body -------------------------------------------------------------------
    p0 = dyn["pose"][w, 0, 0:3, 3].copy()
    A = np.zeros((nb, 6, 6)); Mg = np.zeros((nb, 6, 6)); Wg = np.zeros((nb, 6))
    for b in range(nb):
        R = dyn["pose"][w, b, 0:3, 0:3]; p = dyn["pose"][w, b, 0:3, 3] - p0
        Mb = m.mass[b]
        M11, M12, M22 = R @ Mb[0:3, 0:3] @ R.T, R @ Mb[0:3, 3:6] @ R.T, R @ Mb[3:6, 3:6] @ R.T
        G22 = M22
        G12 = M12 + np.array([cross(p, M22[:, j]) for j in range(3)]).T               # + P M'22
        G21 = G12.T
        G11 = M11 - np.array([cross(M12[i], p) for i in range(3)]) + np.array([cross(p, G21[:, j]) for j in range(3)]).T
        G = np.block([[G11, G12], [G21, G22]])
        Mg[b] = G
        mm = Mb[3, 3]
        c = np.zeros(3) if mm <= 1e-10 else np.array([Mb[2, 4], Mb[0, 5], Mb[1, 3]]) / mm
        tw = dyn["twist"][w, b]
        Tw = R @ tw[0:3]; Tv = R @ cross(c, tw[0:3]) + cross(p, Tw)                   # T* in world axes
        ow = R @ Om[b, 0:3]; ov = R @ Om[b, 3:6] + cross(p, ow)                         # Om in world axes
        Ab = G / dt
        for j in range(6):                                                              # (-ad(T*)^T) Mg
            Ab[0:3, j] += cross(Tw, G[0:3, j]) + cross(Tv, G[3:6, j])
            Ab[3:6, j] += cross(Tw, G[3:6, j])
        for r in range(6):                                                              # Mg ad(Om)
            Ab[r, 0:3] += cross(G[r, 0:3], ow) + cross(G[r, 3:6], ov)
            Ab[r, 3:6] += cross(G[r, 3:6], ow)
        if np.any(m.visc[b] != 0):
            Ai = O.iadjoint(np.block([[R, p[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]))
            Ab += Ai.T @ m.visc[b] @ Ai
        A[b] = Ab
        f = R @ PT[b, 3:6]
        Wg[b] = np.concatenate([R @ PT[b, 0:3] + cross(p, f), f])
    # subtree sums
    Ac, Mc, Wc = A.copy(), Mg.copy(), Wg.copy()
    for b in range(nb - 1, 0, -1):
        p = int(m.parent[b])
        if p >= 0:
            Ac[p] += Ac[b]; Mc[p] += Mc[b]; Wc[p] += Wc[b]
    # ---- B2.2 lane = dof ---------------------------------------------------------------------
    X = np.zeros((n, 6)); dX = np.zeros((n, 6)); P = np.zeros((n, 6)); Rr = np.zeros((n, 6)); Gk = np.zeros((n, 6)); rhs = np.zeros(n)
    for k in range(n):
        b = body_of[k]
        R = dyn["pose"][w, b, 0:3, 0:3]; p = dyn["pose"][w, b, 0:3, 3] - p0
        S, dS = dyn["jac"][w, b, :, k], dyn["djac"][w, b, :, k]
        xw = R @ S[0:3]; X[k] = np.concatenate([xw, R @ S[3:6] + cross(p, xw)])
        aw = dS[0:3] - cross(Om[b, 0:3], S[0:3])
        av = dS[3:6] - cross(Om[b, 3:6], S[0:3]) - cross(Om[b, 0:3], S[3:6])
        dw = R @ aw; dX[k] = np.concatenate([dw, R @ av + cross(p, dw)])
        P[k] = Ac[b].T @ X[k]; Rr[k] = Mc[b] @ X[k]; Gk[k] = Ac[b] @ X[k] + Mc[b] @ dX[k]
        rhs[k] = X[k] @ Wc[b]
    # masks
    anc = np.zeros((nb, n), bool)
    for b in range(nb):
        a = b
        while a >= 0:
            anc[b, m.dof_off[a]:m.dof_off[a] + m.jnd[a]] = True
            a = int(m.parent[a])
    up = anc[body_of]                                  # up[k, i]: body(i) ancestor-or-equal of body(k)
    desc = up.T & ~up                                  # desc[k, i]: body(i) strict descendant of body(k)
    Z = np.zeros((n, n))
    for i in range(n):
        for k in range(n):
            if up[k, i]:
                Z[i, k] = X[i] @ Gk[k]
            elif desc[k, i]:
                Z[i, k] = P[i] @ X[k] + Rr[i] @ dX[k]
    print(name, "world", w, "Z err %.2e" % (np.abs(Z - Zref[w]).max() / np.abs(Zref[w]).max()),
          " rhs err %.2e" % (np.abs(rhs - rhs_ref[w]).max() / max(1, np.abs(rhs_ref[w]).max())))
    # ---- constraint rows in world form ---------------------------------------------------------
    if m.nc:
        gf, cf, info = O.update_constraints(m, dyn, q[w:w + 1], dq[w:w + 1], dt, gforce[w:w + 1], Y[w:w + 1])
        # info does not return jacs: recompute the reference rows here from body Jacobians
        for c in range(m.nc):
            if m.ctype[c] != O.CT_SOFTFINGER:
                continue
            b0, b1 = int(m.c_body0[c]), int(m.c_body[c])
            Hc0 = info["frames0"][0, c].copy()
            J1 = O.iadjoint(O.hinv(dyn["pose"][w, b1]) @ Hc0) @ dyn["jac"][w, b1] if b1 >= 0 else np.zeros((6, n))
            J0 = O.iadjoint(O.hinv(dyn["pose"][w, b0]) @ Hc0) @ dyn["jac"][w, b0] if b0 >= 0 else np.zeros((6, n))
            Jref = (J1 - J0)[2:6]
            Rc = Hc0[0:3, 0:3]; r = Hc0[0:3, 3] - p0
            Rx = Rc.T; px = -Rc.T @ r
            rows = np.zeros((4, n))
            for k in range(n):
                s = (1. if (b1 >= 0 and anc[b1, k]) else 0.) - (1. if (b0 >= 0 and anc[b0, k]) else 0.)
                cw = Rx @ X[k, 0:3]; cv = Rx @ X[k, 3:6] + cross(px, cw)
                rows[:, k] = s * np.array([cw[2], cv[0], cv[1], cv[2]])
            print("   contact %d rows err %.2e" % (c, np.abs(rows - Jref).max()))
