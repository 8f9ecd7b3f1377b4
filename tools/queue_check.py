#!/usr/bin/env python3
"""Work queue against one workgroup per world: bitwise equality of whole episodes and timing (GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
for nc, B in ((4, 4096), (4, 6000), (8, 4096), (0, 8192)):
    m = scenes.flat(scenes.human36_world(nc))
    bw = BatchedWorlds(m)
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    res = {}
    for static in (True, False):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32) if nc else None
        bw.step(tq, tdq, 5e-3, 40, cforce=cf, static_worlds=static)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            tq2, tdq2 = bw.to_device(q, dq, torch.float32)
            cf2 = bw.new_cforce(B, torch.float32) if nc else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            bw.step(tq2, tdq2, 5e-3, 40, cforce=cf2, static_worlds=static)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[static] = (tq, tdq, cf, min(ts))
    same = torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1]) and \
        (nc == 0 or torch.equal(res[True][2], res[False][2]))
    print("nc %d B %d: bitwise equal %s, finite %s; episode launch static %.3f ms, queue %.3f ms (%.1f -> %.1f M world-steps/s)"
          % (nc, B, same, bool(torch.isfinite(res[False][0]).all()), res[True][3] * 1e3, res[False][3] * 1e3,
             B * 40 / res[True][3] / 1e6, B * 40 / res[False][3] / 1e6), flush=True)
    assert same
    bw.close()
print("queue ok")
