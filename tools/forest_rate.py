#!/usr/bin/env python3
"""Rate of a small model through the library's forest for a range of batch sizes and build pins (GPU box).
usage: python tools/forest_rate.py [model] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_model
from arboris_python_amd.batch import BatchedWorlds
name = sys.argv[1] if len(sys.argv) > 1 else "simplearm"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m, q0, dq0 = load_model(name)
bw = BatchedWorlds(m)
print(name, "forest copies", bw.info["forest_copies"])
rng = np.random.default_rng(3)
for B in (4096, 16384, 65536):
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + 0.2 * rng.standard_normal((B, m.ndof))
    lin = m.dof2q >= 0
    q[:, m.dof2q[lin]] += 0.2 * rng.standard_normal((B, int(lin.sum())))
    for dtype in (torch.float32, torch.float64):
        tq, tdq = bw.to_device(q, dq, dtype)
        line = "%7d worlds %s:" % (B, "f32" if dtype == torch.float32 else "f64")
        for label, kw in (("auto", {}), ("w2", dict(waves=2)), ("w3", dict(waves=3)), ("one world", dict(one_world=True)),
                          ("one world w2", dict(one_world=True, waves=2)), ("one world w3", dict(one_world=True, waves=3))):
            cf = bw.new_cforce(B, dtype) if m.nc else None
            ts = []
            for _ in range(4):
                a, b = tq.clone(), tdq.clone()
                if cf is not None: cf.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                bw.step(a, b, 1e-3, T, cforce=cf, **kw)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            p = bw.plan(B, T, dtype=dtype, **kw)
            line += "  %s %.0f M (%dw x%d, %d B)" % (label, B * T / min(ts) / 1e6, p["waves_per_simd"], p["worlds_per_wavefront"], p["lds_bytes"])
        print(line, flush=True)
bw.close()
