#!/usr/bin/env python3
"""Error distribution of the float32 fused kernel against the float64 oracle over many (world, step) pairs of
the falling episode: every sampled state logged by the device is stepped once by the oracle, and every pair over the
1e-5 gate goes through tests/parity_tools.explain_outlier (a decision difference only counts when the decision was
marginal for the oracle itself) or the ill-conditioning rule of the tests: explained / unexplained counts are printed.
usage (GPU box): python tools/replay_stats.py [seed [world_stride [step_stride [contacts]]]]   (defaults 1000, 16, 3, 4)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
import arb_oracle as O
from conftest import load_model, oracle_sensitivity
from parity_tools import explain_outlier, ill_conditioned
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
m, _, _ = load_model("human36_c%d" % arg(4, 4))
bw = BatchedWorlds(m)
B, T, dt = 4096, 40, 5e-3
q, dq = synth.standing_states(m, B, seed=arg(1, 1000), drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), log_energy=False)
torch.cuda.synchronize()
worlds = np.arange(0, B, arg(2, 16))     # 256 worlds by default
errs, reasons, unexplained = [], {}, []
for k in range(0, T - 1, arg(3, 3)):       # 13 steps by default
    qk = log["q"][k][worlds].double().cpu().numpy(); dqk = log["dq"][k][worlds].double().cpu().numpy()
    oq, odq, _ = O.step(m, qk, dqk, dt)
    g = log["dq"][k + 1][worlds].double().cpu().numpy()
    gq = log["q"][k + 1][worlds].double().cpu().numpy()
    e = np.abs(g - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
    eq = np.abs(gq - oq).max(axis=1) / np.maximum(1., np.abs(oq).max(axis=1))
    errs.append(np.maximum(e, eq))
    for i in np.flatnonzero((e > 1e-5) | (eq > 1e-5)):
        w = int(worlds[i])
        qf, dqf = log["q"][k][w].cpu().numpy(), log["dq"][k][w].cpu().numpy()
        why = explain_outlier(bw, m, qf, dqf, dt)
        if why is None:
            why = ill_conditioned(m, qf, dqf, dt, eq[i], e[i])
        tag = "UNEXPLAINED" if why is None else why.split(":")[0].split("(")[0].strip()
        reasons[tag] = reasons.get(tag, 0) + 1
        if why is None:
            unexplained.append((k, w, float(eq[i]), float(e[i])))
        print("  outlier step %2d world %4d: err q %.2e dq %.2e -- %s" % (k, w, eq[i], e[i], why))
    print("step %2d: median %.1e  p99 %.1e  max %.1e  > 1e-5: %d of %d" % (k, np.median(e), np.quantile(e, 0.99), e.max(), int((e > 1e-5).sum()), len(e)))
    sys.stdout.flush()
e = np.concatenate(errs)
print("all: %d pairs, within 1e-5: %.2f %%, within 1e-4: %.2f %%, max %.1e" % (len(e), 100 * (e <= 1e-5).mean(), 100 * (e <= 1e-4).mean(), e.max()))
print("outliers: %d, by reason: %s" % (int((e > 1e-5).sum()), reasons))
print("unexplained: %d %s" % (len(unexplained), unexplained[:20]))
