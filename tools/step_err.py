#!/usr/bin/env python3
"""One-step float32 errors against the oracle on states harvested from a falling episode, per build (development).
usage: step_err.py [--contacts 8] name ...   (names: builds under build/ab, or shipped)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import arb_oracle as O
from arboris_python_amd import scenes, synth, _capi
from arboris_python_amd.batch import BatchedWorlds
args = sys.argv[1:]
nc = 8
if args[0] == "--contacts":
    nc = int(args[1]); args = args[2:]
m = scenes.flat(scenes.human36_world(nc))
B, T = 2048, 40
q, dq = synth.world_states(m, range(B), "standing", 1000, drop=0.03, vel=0.1)
ref = BatchedWorlds(m)
tq, tdq = ref.to_device(q, dq, torch.float32)
log = ref.rollout(tq, tdq, 5e-3, T, cforce=ref.new_cforce(B, torch.float32), log_energy=False)
torch.cuda.synchronize()
worlds = np.arange(3, B, 24)
steps = list(range(4, T, 2))
Q = torch.cat([log["q"][k][worlds] for k in steps]).contiguous()
DQ = torch.cat([log["dq"][k][worlds] for k in steps]).contiguous()
oq, odq, _ = O.step(m, Q.double().cpu().numpy(), DQ.double().cpu().numpy(), 5e-3)
rel = lambda a, b: np.max(np.abs(a - b), axis=1) / np.maximum(1., np.max(np.abs(b), axis=1))
print("%d states" % len(Q))
for n in args:
    path = _capi.LIB_PATH if n == "shipped" else os.path.join(ROOT, "build", "ab", n + ".so")
    bw = BatchedWorlds(m, lib=_capi._open(path))
    for gk in ((False, True) if n == "shipped" else (False,)):
        a, b = Q.clone(), DQ.clone()
        bw.step(a, b, 5e-3, 1, cforce=bw.new_cforce(len(Q), torch.float32), general_kernels=gk)
        torch.cuda.synchronize()
        e = np.maximum(rel(a.double().cpu().numpy(), oq), rel(b.double().cpu().numpy(), odq))
        print("%-10s %s feat %2d: share > 1e-5 %.4f  > 2e-5 %.4f  > 1e-4 %.4f   median %.2e p90 %.2e p99 %.2e max %.2e"
              % (n, "general" if gk else "default", bw.plan(len(Q), 1, general_kernels=gk)["feat"], (e > 1e-5).mean(), (e > 2e-5).mean(), (e > 1e-4).mean(),
                 np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max()))
    bw.close()
