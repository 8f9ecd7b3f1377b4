#!/usr/bin/env python3
"""Final states of a few fixed workloads computed with the library ARBSTEP_LIB points at (default: the built one), saved
to an .npz: two builds are compared bit for bit with tools/xlib_cmp.py (development: refactorings that must not change
a single bit).  usage (GPU box): ARBSTEP_LIB=build/ab/x.so python tools/xlib_dump.py out.npz [quick]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
out = {}
cases = [("human36_c4", 700, 40, "standing"), ("human36_c4", 5000, 24, "standing"), ("human36_g", 512, 8, "random")]
if not quick:
    cases += [("human36_c8", 600, 24, "standing"), ("human36_c8", 3000, 12, "standing")]
for name, B, T, kind in cases:
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    kw = dict(drop=0.03, vel=0.1) if kind == "standing" else {}
    q, dq = synth.world_states(m, range(B), kind, 31, **kw)
    if kind == "standing":
        q[:, 7] -= 0.015
    for dtype in ([torch.float32] if quick else [torch.float32, torch.float64]):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype) if m.nc else None
        bw.step(tq, tdq, 5e-3, T, cforce=cf)
        torch.cuda.synchronize()
        bw.status()
        key = "%s_%d_%d_%s" % (name, B, T, str(dtype).split(".")[-1])
        out[key + "_q"] = tq.cpu().numpy(); out[key + "_dq"] = tdq.cpu().numpy()
        if cf is not None:
            out[key + "_cf"] = cf.cpu().numpy()
        if m.nc:
            # user torques: the FEAT 1 kernel
            tau = torch.as_tensor(np.random.default_rng(5).uniform(-0.05, 0.05, size=(B, m.ndof)), dtype=dtype, device=bw.device)
            tau[:, :6] = 0.
            tq, tdq = bw.to_device(q, dq, dtype)
            cf = bw.new_cforce(B, dtype)
            bw.step(tq, tdq, 5e-3, min(T, 12), cforce=cf, ext_gforce=tau.contiguous())
            torch.cuda.synchronize()
            out[key + "_ext_q"] = tq.cpu().numpy(); out[key + "_ext_dq"] = tdq.cpu().numpy()
    bw.close()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], len(out), "arrays")
