#!/usr/bin/env python3
"""Time the fused and the split execution of the contact workload at several batch sizes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = scenes.flat(scenes.human36_world(nc))
bw = BatchedWorlds(m)
sizes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1024, 2048, 4096, 8192, 16384, 65536]
for B in sizes:
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    q0, dq0 = bw.to_device(q, dq, torch.float32)
    res = []
    for fused in (True, False):
        tq, tdq = q0.clone(), dq0.clone()
        cf = bw.new_cforce(B, torch.float32)
        bw.step(tq, tdq, 5e-3, 2, cforce=cf, fused=fused)
        torch.cuda.synchronize()
        tq.copy_(q0); tdq.copy_(dq0)
        t0 = time.perf_counter()
        bw.step(tq, tdq, 5e-3, 40, cforce=cf, fused=fused)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 40 * 1e3)
    sys.stdout.flush()
    print("B=%6d  fused %.3f ms/step (%.2f M steps/s)   split %.3f ms/step (%.2f M steps/s)"
          % (B, res[0], B / res[0] / 1e3, res[1], B / res[1] / 1e3))
