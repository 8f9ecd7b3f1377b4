#!/usr/bin/env python3
"""Split execution at a large batch with 1, 2 and 4 worlds per wavefront in the sweep kernel (ARB_GSW_PACK=0|2|4): world-steps/s
of the whole split launch sequence, against the fused default.  usage (GPU box): python tools/gsn_time.py [batch [contacts]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = scenes.flat(scenes.human36_world(nc))
bw = BatchedWorlds(m)
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
T = 40
for mode, pack, wv in (("fused", "0", "3"), ("split, 1 world / wavefront", "0", "3"), ("split, 2 worlds / wavefront", "2", "3"),
                       ("split, 4 worlds / wavefront", "4", "3"), ("split, 2 worlds / wavefront, 256 regs", "2", "2"),
                       ("split, 4 worlds / wavefront, 256 regs", "4", "2")):
    os.environ["ARB_GSW_PACK"] = pack; os.environ["ARB_GSW_WAVES"] = wv
    best = None
    for rep in range(3):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bw.step(tq, tdq, 5e-3, T, cforce=cf, split=("wave" if mode != "fused" else False))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    print("%-40s %7.3f M world-steps/s  (%.2f ms per episode)" % (mode, B * T / best / 1e6, best * 1e3), flush=True)
