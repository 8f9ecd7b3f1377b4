#!/usr/bin/env python3
"""Pivot growth (arb_inspect_out.pivot_growth, the quantity behind ARB_WARN_ILLCOND) along episodes of the bench workloads."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
for nc, kind, kw in ((0, "random", dict(angle=0.7, vel=1.0)), (4, "standing", dict(drop=0.03, vel=0.1)), (8, "standing", dict(drop=0.03, vel=0.1))):
    m = scenes.flat(scenes.human36_world(nc))
    bw = BatchedWorlds(m)
    B = 5000
    q, dq = synth.world_states(m, range(B), kind, 3, **kw)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32) if nc else None
    for k in range(41):
        g = bw.inspect(tq, tdq, 5e-3, ["pivot_growth"], cforce=cf)["pivot_growth"]
        if k % 5 == 0:
            qs = torch.quantile(g.double(), torch.tensor([0.5, 0.99, 0.9999, 1.0], dtype=torch.float64, device=g.device)).tolist()
            print("nc %d step %2d growth p50 %.1f p99 %.1f p99.99 %.1f max %.1f   max|dq| %.1f  nonfinite %d"
                  % (nc, k, qs[0], qs[1], qs[2], qs[3], float(tdq.abs().max()), int((~torch.isfinite(tdq)).any(dim=1).sum())), flush=True)
        bw.step(tq, tdq, 5e-3, 1, cforce=cf)
    torch.cuda.synchronize()
    print("warnings", bw.warnings())
    bw.close()
