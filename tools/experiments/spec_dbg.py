#!/usr/bin/env python3
"""Development: specialised against general kernels (ARB_FORCE_SPEC) on one model, step by step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
name = sys.argv[1] if len(sys.argv) > 1 else "human36_c8"
dt_ = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
m, _, _ = load_model(name)
bw = BatchedWorlds(m)
B = 256
q, dq = synth.standing_states(m, B, seed=5, drop=0.03, vel=0.1)
q[:, 7] -= 0.025
st = {}
for key in ("spec", "general"):
    os.environ["ARB_FORCE_SPEC"] = "1" if key == "spec" else "0"
    print(key, bw.plan(B, 1, dtype=dt_))
    tq, tdq = bw.to_device(q, dq, dt_)
    cf = bw.new_cforce(B, dt_)
    hist = []
    for k in range(12):
        bw.step(tq, tdq, 5e-3, 1, cforce=cf)
        torch.cuda.synchronize()
        hist.append((tq.clone(), tdq.clone(), cf.clone()))
    st[key] = hist
for k in range(12):
    a, b = st["spec"][k], st["general"][k]
    print(k, [float((x - y).abs().max()) for x, y in zip(a, b)], "max|cf| spec %.3g general %.3g" % (float(a[2].abs().max()), float(b[2].abs().max())),
          "active rows spec %d general %d" % (int((a[2] != 0).sum()), int((b[2] != 0).sum())))
a, b = st["spec"][0], st["general"][0]
ca, cb = a[2].cpu().numpy().reshape(B, -1, 4), b[2].cpu().numpy().reshape(B, -1, 4)
print("per-contact nonzero worlds spec   ", (np.abs(ca).max(axis=2) > 0).sum(axis=0))
print("per-contact nonzero worlds general", (np.abs(cb).max(axis=2) > 0).sum(axis=0))
print("world 0 spec\n", ca[0], "\nworld 0 general\n", cb[0])
