#!/usr/bin/env python3
"""Cycles of the segments of a sliding solve (development build -DARB_GSSTAMPS, loaded through ARBSTEP_LIB), summed
per world over one step's sliding solves: [0] top of the contact iteration .. release/static decision, [1] .. c1/kappa
and the block's constants, [2] .. leftmost root, [3] .. 4x4 solve, [4] .. force hand-over and velocity update."""
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
for k in range(40):
    if k in (20, 28, 36, 39):
        r = bw.inspect(tq, tdq, 5e-3, ["gs_stats", "stamps"], cforce=cf)
        st = r["stamps"].double()
        cnt = st[:, 5].clamp(min=1)
        per = (st[:, :5] / cnt[:, None])[st[:, 5] > 0].mean(0).tolist()
        print("step %d: sliding solves/world %.1f; cycles per sliding solve: decide %.0f  c1/kappa %.0f  root %.0f  solve4 %.0f  tail %.0f  (sum %.0f)"
              % (k, float(st[:, 5].mean()), *per, sum(per)))
        print("         warm start certified in %.1f %% of the sliding solves; Laguerre iterations per solve %.2f"
              % (100 * float(st[:, 6].sum() / st[:, 5].sum()), float(st[:, 7].sum() / st[:, 5].sum())))
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
