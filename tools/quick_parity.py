#!/usr/bin/env python3
"""Single-step float32 parity of a development library (ARBSTEP_LIB, -DARB_QUICK builds: human36, float32)
against the oracle: random states without contacts, and standing states pressed into the floor."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import arb_oracle as O
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
m = scenes.flat(scenes.human36_world(4))
bw = BatchedWorlds(m, 0)
for name, (q, dq) in (("random, in the air", synth.random_states(m, 64, seed=3)),
                      ("standing, 1 cm into the floor", synth.standing_states(m, 64, seed=1, drop=0.03, vel=0.1))):
    if name.startswith("standing"):
        q[:, 7] -= 0.01
    else:
        q[:, 7] += 2.0
    q32 = q.astype(np.float32).astype(np.float64); dq32 = dq.astype(np.float32).astype(np.float64)
    oq, odq, _ = O.step(m, q32, dq32, 5e-3)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(64, torch.float32)
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
    torch.cuda.synchronize()
    eq = np.abs(tq.cpu().numpy() - oq).max() / max(1., np.abs(oq).max())
    edq = np.abs(tdq.cpu().numpy() - odq).max() / max(1., np.abs(odq).max())
    print("%-32s rel err q %.2e dq %.2e   max contact force %.0f" % (name, eq, edq, float(cf.abs().max())))
