#!/usr/bin/env python3
"""Split execution pipelined over K chunks on K streams (K model handles), against the fused kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = scenes.flat(scenes.human36_world(nc))
KS = [1, 2, 4, 8]
bws = [BatchedWorlds(m) for _ in range(max(KS))]
streams = [torch.cuda.Stream() for _ in range(max(KS))]
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
q0, dq0 = bws[0].to_device(q, dq, torch.float32)


def run(K, split, nsteps=40):
    tq, tdq = q0.clone(), dq0.clone()
    cf = bws[0].new_cforce(B, torch.float32)
    ch = B // K
    parts = [(tq[i * ch:(i + 1) * ch], tdq[i * ch:(i + 1) * ch], cf[i * ch:(i + 1) * ch]) for i in range(K)]
    out = None
    for rep in range(2):
        tq.copy_(q0); tdq.copy_(dq0); cf.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, (a, b, c) in enumerate(parts):
            bws[i].step(a, b, 5e-3, nsteps, cforce=c, stream=streams[i], fused=not split, split=split)
        torch.cuda.synchronize()
        out = (time.perf_counter() - t0) / nsteps * 1e3
    return out, tq.clone()


base, ref = run(1, False)
print("B=%d nc=%d fused K=1: %.3f ms/step (%.2f M/s)" % (B, nc, base, B / base / 1e3))
for K in KS:
    for split in (False, True):
        if K == 1 and not split:
            continue
        t, res = run(K, split)
        print("  K=%d %-5s %.3f ms/step (%.2f M/s)   max|dq - fused| %.2e"
              % (K, "split" if split else "fused", t, B / t / 1e3, float((res - ref).abs().max())))
