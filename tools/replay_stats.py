#!/usr/bin/env python3
"""Error distribution of the float32 fused kernel against the float64 oracle over many (world, step) pairs of
the falling episode: every sampled state logged by the device is stepped once by the oracle.
usage (GPU box): python tools/replay_stats.py [seed [world_stride [step_stride]]]   (defaults 1000, 16, 3)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
import arb_oracle as O
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
B, T, dt = 4096, 40, 5e-3
q, dq = synth.standing_states(m, B, seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), log_energy=False)
torch.cuda.synchronize()
worlds = np.arange(0, B, int(sys.argv[2]) if len(sys.argv) > 2 else 16)     # 256 worlds by default
errs = []
for k in range(0, T - 1, int(sys.argv[3]) if len(sys.argv) > 3 else 3):       # 13 steps by default
    qk = log["q"][k][worlds].double().cpu().numpy(); dqk = log["dq"][k][worlds].double().cpu().numpy()
    oq, odq, _ = O.step(m, qk, dqk, dt)
    g = log["dq"][k + 1][worlds].double().cpu().numpy()
    e = np.abs(g - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
    errs.append(e)
    print("step %2d: median %.1e  p99 %.1e  max %.1e  > 1e-5: %d of %d" % (k, np.median(e), np.quantile(e, 0.99), e.max(), int((e > 1e-5).sum()), len(e)))
    sys.stdout.flush()
e = np.concatenate(errs)
print("all: %d pairs, within 1e-5: %.2f %%, within 1e-4: %.2f %%, max %.1e" % (len(e), 100 * (e <= 1e-5).mean(), 100 * (e <= 1e-4).mean(), e.max()))
