#!/usr/bin/env python3
"""Development: the packed sweep kernel (two worlds per wavefront, ARB_GSW_PACK=1) against the one-world sweep kernel of
the split execution and against the fused kernel, bit for bit.  usage (GPU box): python tools/gsw_pack_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
bad = 0
for name, B, T in (("human36_c4", 1001, 40), ("human36_c8", 600, 40), ("human36_c4", 3, 40)):
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    q, dq = synth.world_states(m, range(B), "standing", 11, drop=0.03, vel=0.2)
    q[:, 7] -= 0.01
    for dtype in (torch.float32, torch.float64):
        res = {}
        for mode in ("fused", "wave", "pack"):
            os.environ["ARB_GSW_PACK"] = "1" if mode == "pack" else "0"
            tq, tdq = bw.to_device(q, dq, dtype)
            cf = bw.new_cforce(B, dtype)
            bw.step(tq, tdq, 5e-3, T, cforce=cf, split=("wave" if mode != "fused" else False), waves=2)
            torch.cuda.synchronize()
            res[mode] = (tq, tdq, cf)
        for mode in ("wave", "pack"):
            same = all(torch.equal(a, b) for a, b in zip(res["fused"], res[mode]))
            d = max(float((a - b).abs().max()) for a, b in zip(res["fused"], res[mode]))
            nd = int((res["fused"][1] != res[mode][1]).any(dim=1).sum())
            print("%s B=%d %s: %s vs fused: %s (max diff %.2e, %d worlds differ), max force %.1f" % (name, B, dtype, mode, "IDENTICAL" if same else "DIFFER", d, nd, float(res[mode][2].abs().max())))
            bad += 0 if same else 1
    bw.close()
print("mismatches:", bad)
