#!/usr/bin/env python3
"""Development: do the worlds that take the 6x6 eigenvalue fallback at step k take it again at step k + 1?
(one-step launches: a longest-first order needs a predictor)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
B, T = 4096, 40
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
prev = None
for k in range(T):
    st = bw.inspect(tq, tdq, 5e-3, ["gs_stats"], cforce=cf.clone())["gs_stats"].cpu().numpy()
    slow = set(np.nonzero(st[:, 3] > 0)[0].tolist())
    nslide = st[:, 2] + st[:, 3]
    if k >= 15:
        print(k, "fallback worlds", len(slow), "fallbacks", int(st[:, 3].sum()), "max per world", int(st[:, 3].max()),
              "again from previous step", len(slow & prev) if prev is not None else None,
              "sliding solves median %d p99 %d max %d" % (np.median(nslide), np.percentile(nslide, 99), nslide.max()))
    prev = slow
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
torch.cuda.synchronize()
