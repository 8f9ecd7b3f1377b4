#!/usr/bin/env python3
"""Same-process A/B on the headline workload: the default kernels against ARB_STEP_BODY_COLUMNS (and the general kernels),
interleaved rounds of whole 40-step episodes.  usage (GPU box): python tools/ab_bodycols.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = scenes.flat(scenes.human36_world(4))
bw = BatchedWorlds(m)
q, dq = synth.world_states(m, range(B), "standing", 1000, drop=0.03, vel=0.1)
q0, dq0 = bw.to_device(q, dq, torch.float32)
variants = {"default": {}, "body_columns": dict(body_columns=True), "general": dict(general_kernels=True)}
res = {k: [] for k in variants}
for rnd in range(6):
    for name, kw in variants.items():
        tq, tdq, cf = q0.clone(), dq0.clone(), bw.new_cforce(B, torch.float32)
        n = 60
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            tq.copy_(q0); tdq.copy_(dq0); cf.zero_()
            bw.step(tq, tdq, 5e-3, 40, cforce=cf, **kw)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        if rnd:
            res[name].append(B * 40 * n / el / 1e6)
for name, v in res.items():
    print("%-13s %s  median %.3f M world-steps/s" % (name, " ".join("%.2f" % x for x in v), float(np.median(v))))
