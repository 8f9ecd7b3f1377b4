#!/usr/bin/env python3
"""Sub-phase cycles of phase A (library built with -DARB_ASTAMPS -DARB_QUICK -DARB_QUICK_INSPECT; lane 0 = the root body)."""
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
names = ["joint local + H_pc", "block algebra", "own columns", "level loop", "body wrenches", "qd copy"]
for k in range(13):
    if k in (0, 12):
        r = bw.inspect(tq, tdq, 5e-3, ["stamps"])
        ph = (r["stamps"][:, 1:7] - r["stamps"][:, 0:6]).double().mean(0).tolist()
        print("step %2d  " % k + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, ph)))
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
