#!/usr/bin/env python3
"""Would a float32 BODY-FRAME composite recursion (leaf to root, no world-frame prefix sums) be accurate enough for phase B?
(rounds 3 and 4 reviews; NumPy experiment on the CPU, float64 oracle as the judge -- test infrastructure, not product code)

The kernel assembles Z = M/dt + B + N in float64 (world-frame blocks, subtree sums as differences of prefix sums) and rounds
it to float32 once: every entry of the float32 kernels' Z is the correctly rounded one.  The alternative evaluated here: the
composite-rigid-body recursion in the bodies' own frames,  I_c(b) = M_b + sum_children X_c^T I_c(c) X_c,  F = I_c(b) S_b,
walked up the ancestors -- every operation in float32, on joint transforms computed in float64 and rounded to float32 (the
favourable case), B + N taken from the float64 assembly (favourable again: with dt = 5 ms M/dt is ~200 x the rest).

For states of a falling episode (64 worlds x 40 steps, the bench's distribution) it prints the error of Z and -- Z pushed
through the float64 reference step, so that ONLY the assembly differs -- the error of the step's new velocities against the
1e-5 gate, for (a) float64 assembly rounded to float32 (today's kernel input) and (b) the float32 body-frame recursion.

usage: python tests/experiments/bodyframe_f32_assembly.py [contacts=4] [worlds=64]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conftest import load_model                      # noqa: E402
from oracle import arb_oracle as O                    # noqa: E402
from arboris_python_amd import synth                  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m, _, _ = load_model("human36_c%d" % nc)
dt = 5e-3
f32 = np.float32


def crba_f32(m, q, dq):
    """M by the body-frame composite recursion, float32 throughout (inputs: float64 joint transforms rounded once)."""
    Bn, n, nb = q.shape[0], m.ndof, m.nb
    X = np.zeros((Bn, nb, 6, 6), f32)             # parent -> child twist transform  Ad(H_pc^-1)
    S = [None] * nb                               # joint columns in the child frame, (B, 6, nd)
    for b in range(nb):
        ds = slice(int(m.dof_off[b]), int(m.dof_off[b] + m.jnd[b]))
        qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
        jt = int(m.jtype[b])
        H_rn, J_nr, _ = O.joint_kinematics(jt, q[:, qs], dq[:, ds])
        H_pc = m.H_pr[b] @ (H_rn @ O.hinv(m.H_cn[b]))
        X[:, b] = O.iadjoint(H_pc).astype(f32)
        S[b] = (O.adjoint(m.H_cn[b]) @ J_nr).astype(f32)
    Ic = np.broadcast_to(m.mass.astype(f32), (Bn, nb, 6, 6)).copy()
    for b in range(nb - 1, -1, -1):               # DFS preorder: children come after their parent
        p = int(m.parent[b])
        if p >= 0:
            Xt = np.swapaxes(X[:, b], -1, -2)
            Ic[:, p] = Ic[:, p] + (Xt @ Ic[:, b]) @ X[:, b]
    M = np.zeros((Bn, n, n), f32)
    for b in range(nb):
        nd = int(m.jnd[b])
        if nd == 0:
            continue
        ds = slice(int(m.dof_off[b]), int(m.dof_off[b]) + nd)
        F = Ic[:, b] @ S[b]                        # (B, 6, nd)
        M[:, ds, ds] = np.swapaxes(S[b], -1, -2) @ F
        j = b
        while int(m.parent[j]) >= 0:
            F = np.swapaxes(X[:, j], -1, -2) @ F
            j = int(m.parent[j])
            if int(m.jnd[j]) == 0:
                continue
            dj = slice(int(m.dof_off[j]), int(m.dof_off[j] + m.jnd[j]))
            blk = np.swapaxes(S[j], -1, -2) @ F   # (B, ndj, nd)
            M[:, dj, ds] = blk
            M[:, ds, dj] = np.swapaxes(blk, -1, -2)
    assert M.dtype == f32
    return M


def step_with(m, dyn, Mrepl, q, dq, cf):
    d = dict(dyn)
    d["M"] = Mrepl
    gforce, Z, Y = O.update_controllers(m, d, q, dq, dt)
    gtot, cfn, _ = O.update_constraints(m, d, q, dq, dt, gforce, Y, cf.copy())
    qn, dqn = O.integrate(m, d, q, dq, dt, gtot, Y)
    return qn, dqn, cfn


q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
cf = np.zeros((B, m.nc, 4))
eZ = {"a": [], "b": []}
eV = {"a": [], "b": []}
for t in range(40):
    dyn = O.update_dynamic(m, q, dq)
    Z = dyn["M"] / dt + dyn["Bv"] + dyn["N"]
    rest = dyn["Bv"] + dyn["N"]
    Za = Z.astype(f32).astype(np.float64)
    Zb = (crba_f32(m, q, dq) / f32(dt) + rest.astype(f32)).astype(np.float64)
    ref = step_with(m, dyn, dyn["M"], q, dq, cf)
    scale = np.abs(Z).max(axis=(1, 2))
    for key, Zx in (("a", Za), ("b", Zb)):
        eZ[key].append(np.abs(Zx - Z).max(axis=(1, 2)) / scale)
        got = step_with(m, dyn, (Zx - rest) * dt, q, dq, cf)
        eV[key].append(np.abs(got[1] - ref[1]).max(axis=1) / np.maximum(1., np.abs(ref[1]).max(axis=1)))
    q, dq, cf = ref
for key, what in (("a", "float64 assembly rounded to float32 (today)"), ("b", "float32 body-frame recursion")):
    z = np.concatenate(eZ[key]); v = np.concatenate(eV[key])
    print("%-48s  max|dZ|/max|Z|: median %.1e  max %.1e   |  dq+ error: median %.1e  99 %% %.1e  max %.1e  beyond 1e-5: %.2f %%  beyond 1e-6: %.2f %%"
          % (what, np.median(z), z.max(), np.median(v), np.quantile(v, 0.99), v.max(), 100 * (v > 1e-5).mean(), 100 * (v > 1e-6).mean()))
