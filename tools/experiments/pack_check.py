#!/usr/bin/env python3
"""A/B state check for development libraries (ARBSTEP_LIB): steps a batch whose size is not a multiple of 4
through the falling episode and writes the state after 10/20/30/40 steps, or compares it with a file written
by another library.  usage: pack_check.py out.npz [ref.npz]   (driven by tools/ab_bench.sh)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
m = scenes.flat(scenes.human36_world(4))
bw = BatchedWorlds(m)
B = 1027
q, dq = synth.standing_states(m, B, seed=7, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
snaps = []
for k in range(40):
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
    if k in (9, 19, 29, 39):
        snaps.append((tq.cpu().numpy().copy(), tdq.cpu().numpy().copy(), cf.cpu().numpy().copy()))
torch.cuda.synchronize()
np.savez(sys.argv[1], **{"q%d" % i: s[0] for i, s in enumerate(snaps)}, **{"dq%d" % i: s[1] for i, s in enumerate(snaps)},
         **{"cf%d" % i: s[2] for i, s in enumerate(snaps)})
print("finite", bool(np.isfinite(snaps[-1][0]).all()), "max|dq|", float(np.abs(snaps[-1][1]).max()), "max cf", float(np.abs(snaps[-1][2]).max()))
if len(sys.argv) > 2:
    r = np.load(sys.argv[2])
    for i in range(4):
        for k, a in (("q", snaps[i][0]), ("dq", snaps[i][1]), ("cf", snaps[i][2])):
            b = r["%s%d" % (k, i)]
            print("snap %d %s: identical %s  max abs diff %.3e (scale %.3e)" % (i, k, bool(np.array_equal(a, b)), float(np.abs(a - b).max()), float(np.abs(b).max())))
