#!/usr/bin/env python3
"""Cycles inside phase C and D (library built with -DARB_CSTAMPS -DARB_QUICK -DARB_QUICK_INSPECT):
stamps 0 A | 1 A' | 2 B | 3 C: load columns | 4 C: pre-rotation + pivot loop | 5 C: add gvel | 6 D | 7 GS prep .."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
m = scenes.flat(scenes.human36_world(4))
bw = BatchedWorlds(m)
B = 4096
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
names = ["A", "A'", "B", "C load", "C pivots", "C +gvel", "D"]
for k in range(24):
    if k in (0, 12, 23):
        r = bw.inspect(tq, tdq, 5e-3, ["stamps"])
        ph = (r["stamps"][:, 1:] - r["stamps"][:, :-1]).double().mean(0).tolist()
        print("step %2d  " % k + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, ph)))
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
