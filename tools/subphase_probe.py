#!/usr/bin/env python3
"""Sub-phase cycles of a LONE wavefront (one per SIMD: the LDS padded, 4 worlds per CU) on the headline workload, from the
in-kernel stamps of a development build: ARBSTEP_LIB=build/ab/r6_{a,b,c}stamps.so (tools/quick_build.sh with ARB_QUICK=2 and
-DARB_ASTAMPS | -DARB_BSTAMPS | -DARB_CSTAMPS).  usage: ARBSTEP_LIB=... python tools/subphase_probe.py a|b|c"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
which = sys.argv[1]
names = {"a": ["joint local + H_pc", "block algebra", "own columns", "level loop", "body wrenches", "qd copy"],
         "b": ["A", "A'", "B blocks", "B sums", "B dof", "B rows", "B crow"],
         "c": ["A", "A'", "B", "C load", "C pivots", "C +gvel", "D"]}[which]
m = scenes.flat(scenes.human36_world(4))
bw = BatchedWorlds(m)
cus = torch.cuda.get_device_properties(0).multi_processor_count
bw.set_knob("lds_pad", max(0, 32 * 1280 - bw.info["lds_bytes_f32"] - 8))
B = 4 * cus
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
acc = []
for k in range(40):
    r = bw.inspect(tq, tdq, 5e-3, ["stamps"], cforce=cf)
    st = r["stamps"].double()
    d = (st[:, 1:7] - st[:, 0:6]) if which == "a" else (st[:, 1:] - st[:, :-1])
    acc.append(d.mean(0).tolist())
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
a = np.array(acc)
print("lone wavefront, mean over the 40-step episode: " + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, a.mean(0))))
print("   step 0:  " + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, a[0])))
print("   step 35: " + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, a[35])))
