#!/usr/bin/env python3
"""Per-phase shader cycles of one step (s_memtime stamps of the inspect kernel, mean over the batch) for a few models:
A | A' | B | C | D | Gauss-Seidel | E.  usage (GPU box): python tools/phase_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_model
from arboris_python_amd.batch import BatchedWorlds
from arboris_python_amd.flatten import replicate_model
names = ["A", "A'", "B", "C", "D", "GS", "E"]
for name, K, dtype in (("simplearm", 1, torch.float32), ("simplearm", 10, torch.float32), ("simplearm", 1, torch.float64),
                       ("snake9_free_g", 1, torch.float32), ("human36_g", 1, torch.float32), ("human36_c4", 1, torch.float32),
                       ("snake64_g", 1, torch.float64), ("human36_c4", 1, torch.float64), ("human36_c8", 1, torch.float32),
                       ("human36_g", 1, torch.float64)):
    m, q0, dq0 = load_model(name)
    if K > 1:
        m = replicate_model(m, K); q0 = np.tile(q0, K); dq0 = np.tile(dq0, K)
    os.environ["ARB_FOREST"] = "0"
    bw = BatchedWorlds(m)
    B = 4096 if name != "snake64_g" else 2048
    rng = np.random.default_rng(0)
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + 0.1 * rng.standard_normal((B, m.ndof))
    tq, tdq = bw.to_device(q, dq, dtype)
    cf = bw.new_cforce(B, dtype) if m.nc else None
    r = bw.inspect(tq, tdq, 1e-3, ["stamps"], cforce=cf)
    st = r["stamps"].double()
    ph = (st[:, 1:] - st[:, :-1]).mean(0).tolist()
    print("%-14s x%-2d %s (tile %d): " % (name, K, "f32" if dtype == torch.float32 else "f64", bw.info["nmax"]) +
          "  ".join("%s %.0f" % (n, c) for n, c in zip(names, ph)) + "   total %.0f" % sum(ph), flush=True)
    bw.close()
