#!/usr/bin/env python3
"""Would ordering the worlds of a ONE-STEP launch by expected cost (longest first) pay?  Mid-episode states of the
headline workload, physically permuted on the host: by the true cost of the step (sliding solves, from the inspect
kernel's gs_stats), by a proxy available before the launch (contacts with a non-zero force in cforce), or not at all.
(GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from arboris_python_amd.batch import BatchedWorlds
cfg = bench.CONFIGS[3]
m = bench.build_model(cfg)
bw = BatchedWorlds(m)
for B in (4096, 8192):
    q, dq = bench.make_states(cfg, m, 0, B, seed=1000)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    for k0 in (4, 20, 36):
        a, b, c = tq.clone(), tdq.clone(), cf.clone(); c.zero_()
        bw.step(a, b, cfg["dt"], k0, cforce=c)
        torch.cuda.synchronize()
        r = bw.inspect(a, b, cfg["dt"], ["gs_stats"], cforce=c.clone())
        st = r["gs_stats"].cpu().numpy()                 # release, static, sliding fast, sliding slow, sweeps
        true_cost = st[:, 2] * 3 + st[:, 3] * 10 + st[:, 1] + st[:, 0]
        proxy = (c.abs().sum(dim=2) > 0).sum(dim=1).cpu().numpy()
        line = "B %5d step %2d:" % (B, k0)
        for name, key in (("as is", None), ("by true cost", -true_cost), ("by proxy", -proxy), ("worst (ascending)", true_cost)):
            perm = np.arange(B) if key is None else np.argsort(key, kind="stable")
            p = torch.as_tensor(perm, device=bw.device)
            a2, b2, c2 = a[p].contiguous(), b[p].contiguous(), c[p].contiguous()
            ts = []
            for _ in range(5):
                x, y, z = a2.clone(), b2.clone(), c2.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); bw.step(x, y, cfg["dt"], 1, cforce=z); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            line += "  %s %.3f ms" % (name, min(ts))
        print(line, "| mean sliding solves %.1f" % st[:, 2].mean(), flush=True)
bw.close()
