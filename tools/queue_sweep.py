#!/usr/bin/env python3
"""Work-queue item sizes on the headline workload (knobs queue_chunk / queue_tail of arbstep_hooks.h): world-steps/s of whole
40-step episodes, 4096 worlds, float32.  usage (GPU box): python tools/queue_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = scenes.flat(scenes.human36_world(int(os.environ.get("SWEEP_CONTACTS", "4"))))
bw = BatchedWorlds(m)
q, dq = synth.world_states(m, range(B), "standing", 1000, drop=0.03, vel=0.1)
q0, dq0 = bw.to_device(q, dq, torch.float32)
def rate(n=40):
    tq, tdq, cf = q0.clone(), dq0.clone(), bw.new_cforce(B, torch.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tq.copy_(q0); tdq.copy_(dq0); cf.zero_()
        bw.step(tq, tdq, 5e-3, 40, cforce=cf)
    torch.cuda.synchronize()
    return B * 40 * n / (time.perf_counter() - t0) / 1e6
rate(10)
chunks = [int(x) for x in os.environ.get("SWEEP_CHUNKS", "2,3,4,5,6,8,10").split(",")]
tails = [int(x) for x in os.environ.get("SWEEP_TAILS", "0,2,4,6,8").split(",")]
for chunk in chunks:
    row = []
    for tail in tails:
        bw.set_knob("queue_chunk", chunk); bw.set_knob("queue_tail", tail)
        row.append(max(rate(), rate()))
    print("chunk %2d: " % chunk + "  ".join("tail %d: %.2f" % (t, v) for t, v in zip(tails, row)))
