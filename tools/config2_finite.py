#!/usr/bin/env python3
"""bench.py --extra reports finite=false for config 2 (human36 without contacts, random states, 40 steps): which worlds,
from which step, in which precision -- and what does the float64 oracle do with them?  (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
import arb_oracle as O
from arboris_python_amd.batch import BatchedWorlds
cfg = bench.CONFIGS[2]
m = bench.build_model(cfg)
q, dq = bench.make_states(cfg, m, 0, cfg["batch"], seed=0)
bw = BatchedWorlds(m)
for dtype in (torch.float32, torch.float64):
    tq, tdq = bw.to_device(q, dq, dtype)
    first = np.full(len(q), -1)
    big = np.zeros(len(q))
    for k in range(40):
        bw.step(tq, tdq, cfg["dt"], 1)
        torch.cuda.synchronize()
        a = tdq.cpu().numpy()
        bad = ~np.isfinite(a).all(axis=1) | ~np.isfinite(tq.cpu().numpy()).all(axis=1)
        first[(first < 0) & bad] = k
        big = np.maximum(big, np.where(np.isfinite(a).all(axis=1), np.abs(a).max(axis=1), np.inf))
    print(dtype, "non-finite worlds:", np.flatnonzero(first >= 0)[:20], "first step", first[first >= 0][:20],
          "| worlds with |dq| > 1e3 at some step:", int((big > 1e3).sum()), "max finite |dq|", np.nanmax(big[np.isfinite(big)]))
    sel = np.flatnonzero(first >= 0)[:4]
    if len(sel):
        oq, odq = q[sel].copy(), dq[sel].copy()
        for k in range(40):
            oq, odq, _ = O.step(m, oq, odq, cfg["dt"])
            print("  oracle step %2d: max|dq| per world" % k, np.abs(odq).max(axis=1))
            if not np.isfinite(odq).all():
                break
