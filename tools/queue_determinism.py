#!/usr/bin/env python3
"""The work queue schedules (chunk, world) items dynamically: the results must not depend on the schedule.  Repeats the
same launch many times and compares every result with the first, bit for bit (a missing ordering between a world's
hand-over stores and its flag would show up here as a rare mismatch).  usage (GPU box): python tools/queue_determinism.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
for nc, B, T, reps in ((4, 4096, 40, 150), (8, 4096, 40, 60), (4, 65536, 32, 12), (4, 2100, 40, 100)):
    m = scenes.flat(scenes.human36_world(nc)); bw = BatchedWorlds(m)
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    ref = None; bad = 0; t0 = time.perf_counter()
    for r in range(reps):
        tq, tdq = bw.to_device(q, dq, torch.float32); cf = bw.new_cforce(B, torch.float32)
        bw.step(tq, tdq, 5e-3, T, cforce=cf)
        if ref is None:
            ref = (tq, tdq, cf)
            tq2, tdq2 = bw.to_device(q, dq, torch.float32); cf2 = bw.new_cforce(B, torch.float32)
            bw.step(tq2, tdq2, 5e-3, T, cforce=cf2, static_worlds=True)
            assert torch.equal(tq, tq2) and torch.equal(tdq, tdq2) and torch.equal(cf, cf2), "queue != static"
        elif not (torch.equal(tq, ref[0]) and torch.equal(tdq, ref[1]) and torch.equal(cf, ref[2])):
            bad += 1
    torch.cuda.synchronize()
    print("nc %d, %d worlds x %d steps: %d launches, %d differ from the first (%.1f s)" % (nc, B, T, reps, bad, time.perf_counter() - t0), flush=True)
    assert bad == 0
    bw.close()
print("deterministic")
