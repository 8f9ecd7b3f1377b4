#!/usr/bin/env python3
"""Per-phase shader cycles (s_memtime stamps of the inspect kernel, mean over 4096 worlds) of human36 + 4 contacts at steps
0, 12 and 24 of the falling episode: A | A' | B | C | D | Gauss-Seidel | E.  Library: ARBSTEP_LIB (built with ARB_QUICK=2).
usage (GPU box): ARBSTEP_LIB=build/ab/x.so python tools/phase_steps.py"""
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
names = ["A", "A'", "B", "C", "D", "GS", "E"]
for k in range(25):
    if k in (0, 12, 24):
        r = bw.inspect(tq, tdq, 5e-3, ["stamps"], cforce=cf)
        st = r["stamps"].double()
        ph = (st[:, 1:] - st[:, :-1]).mean(0).tolist()
        print("step %2d  " % k + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, ph)) + "   total %.0f" % sum(ph), flush=True)
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
