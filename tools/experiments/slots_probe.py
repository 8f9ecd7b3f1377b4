#!/usr/bin/env python3
"""How many wavefronts of the snake-64 float64 step kernel does a CU hold?  One-step launches of 1024 / 1280 / 1536 worlds:
a batch that fits the wave slots takes one round (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from arboris_python_amd.batch import BatchedWorlds
cfg = bench.CONFIGS[4]
m = bench.build_model(cfg)
bw = BatchedWorlds(m)
print("lds", bw.plan(2048, 1, dtype=torch.float64))
for B in (768, 1024, 1280, 1536, 2048):
    q, dq = bench.make_states(cfg, m, 0, B, seed=1)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    ts = []
    for _ in range(6):
        a, b = tq.clone(), tdq.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); bw.step(a, b, cfg["dt"], 1); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("%5d worlds: %.3f ms" % (B, min(ts)), flush=True)
