#!/usr/bin/env python3
"""Development: the packed step kernel (two worlds per wavefront, ARB_FORCE_PACK=1) against the one-world kernels
(ARB_FORCE_PACK=0), bit for bit: static and queued launches, odd batch sizes, user torques, one launch per step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
bad = 0
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
for B, T, per_step, ext in ((701, 40, False, False), (5001, 24, False, False), (2, 12, False, True), (1, 12, False, False),
                            (333, 12, True, True), (9000, 13, False, True)):
    q, dq = synth.world_states(m, range(B), "standing", 31, drop=0.03, vel=0.2)
    q[:, 7] -= 0.012
    tau = torch.as_tensor(np.random.default_rng(5).uniform(-0.05, 0.05, size=(B, m.ndof)), dtype=torch.float32, device=bw.device)
    tau[:, :6] = 0.
    res = {}
    for mode in ("0", "1"):
        os.environ["ARB_FORCE_PACK"] = mode
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        kw = dict(ext_gforce=tau.contiguous()) if ext else {}
        if per_step:
            for _ in range(T):
                bw.step(tq, tdq, 5e-3, 1, cforce=cf, **kw)
        else:
            bw.step(tq, tdq, 5e-3, T, cforce=cf, **kw)
        torch.cuda.synchronize()
        bw.status()
        res[mode] = (tq, tdq, cf)
    same = all(torch.equal(a, b) for a, b in zip(res["0"], res["1"]))
    nd = int((res["0"][1] != res["1"][1]).any(dim=1).sum())
    print("B=%d T=%d per_step=%s ext=%s: %s (%d worlds differ), finite %s, max force %.1f" % (B, T, per_step, ext, "IDENTICAL" if same else "DIFFER", nd,
          bool(torch.isfinite(res["1"][0]).all()), float(res["1"][2].abs().max())))
    bad += 0 if same else 1
print("mismatches:", bad)
