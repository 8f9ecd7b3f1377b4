#!/usr/bin/env python3
"""How often does the sliding solve leave the register-only root finder for the 6x6 eigenvalue fallback, and do the
worlds that do so explain the slow one-step launches late in the episode?  (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from arboris_python_amd.batch import BatchedWorlds
cfg = bench.CONFIGS[3]
m = bench.build_model(cfg)
bw = BatchedWorlds(m)
B = 4096
q, dq = bench.make_states(cfg, m, 0, B, seed=1000)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
for k in range(40):
    r = bw.inspect(tq, tdq, cfg["dt"], ["gs_stats", "stamps"], cforce=cf.clone())
    st = r["gs_stats"].cpu().numpy()
    sp = r["stamps"].cpu().numpy()
    gs_cyc = (sp[:, 6] - sp[:, 5]).astype(np.float64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); bw.step(tq, tdq, cfg["dt"], 1, cforce=cf); e1.record(); torch.cuda.synchronize()
    slow = st[:, 3]
    if k % 3 == 0 or k > 28:
        print("step %2d: %.3f ms | sliding solves %6d, fallback %4d in %3d worlds | GS cycles: median %6.0f  p99 %7.0f  max %8.0f  (worlds with a fallback: median %7.0f)"
              % (k, e0.elapsed_time(e1), st[:, 2].sum(), slow.sum(), (slow > 0).sum(), np.median(gs_cyc), np.percentile(gs_cyc, 99), gs_cyc.max(),
                 np.median(gs_cyc[slow > 0]) if (slow > 0).any() else 0), flush=True)
bw.close()
