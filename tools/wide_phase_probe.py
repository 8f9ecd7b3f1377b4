#!/usr/bin/env python3
"""Per-phase shader cycles of the wide kernel (inspect stamps): A | A' | B blocks+sums+dofs | Z assembly | C | D + block
inverses | GS | (E is after the last stamp).  usage (GPU box): python tools/wide_phase_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.flatten import flatten_world
from arboris_python_amd.batch import BatchedWorlds
names = ["A", "A'", "B", "Z", "C", "D", "GS"]
def probe(tag, m, q, dq, dt, steps, cf):
    bw = BatchedWorlds(m)
    B = len(q)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    c = bw.new_cforce(B, torch.float64) if cf else None
    for k in range(steps):
        if k in (0, steps - 1):
            r = bw.inspect(tq, tdq, dt, ["stamps"], cforce=c)
            st = r["stamps"].double()
            d = (st[:, 1:] - st[:, :-1]).mean(0).tolist()
            print("%s step %2d: " % (tag, k) + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, d)) + "   sum %.0f cycles" % sum(d))
        bw.step(tq, tdq, dt, 1, cforce=c)
    bw.close()
m = scenes.flat(scenes.snake_world(100))
q, dq = synth.random_states(m, 256, seed=0, angle=0.5, vel=1.0)
probe("snake-100 (256 worlds)", m, q, dq, 1e-3, 2, False)
w = scenes.human36_and_objects_world(4)
m, q0, dq0 = flatten_world(w)
B = 512
probe("human36 + 4 objects (512 worlds)", m, np.tile(q0, (B, 1)), np.tile(dq0, (B, 1)), 5e-3, 30, True)
