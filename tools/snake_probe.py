#!/usr/bin/env python3
"""BASELINE config 4's model (snake-64) through float32 buffers: accuracy against the float64 oracle and throughput of the
float64 kernels, the mixed build (ARB_STEP_MIXED) and plain float32.  usage (GPU box): python tools/snake_probe.py [nlinks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
import arb_oracle as O

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = scenes.flat(scenes.snake_world(nl))
bw = BatchedWorlds(m)
print("snake-%d: info %s" % (nl, bw.info))
dt = 1e-3
q, dq = synth.random_states(m, 2048, seed=0, angle=0.5, vel=1.0)
sub = np.arange(0, 2048, 32)
f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
oq, odq, _ = O.step(m, f32(q[sub]), f32(dq[sub]), dt)
rel = lambda a, b: (np.abs(a - b).max(axis=1) / np.maximum(1., np.abs(b).max(axis=1)))
for name, dtype, kw in (("float64", torch.float64, {}), ("f32 deflt", torch.float32, {}), ("f32 mixed", torch.float32, dict(mixed=True)), ("f32 plain", torch.float32, dict(mixed=False))):
    tq, tdq = bw.to_device(f32(q[sub]), f32(dq[sub]), dtype)
    bw.step(tq, tdq, dt, 1, **kw)
    torch.cuda.synchronize()
    e = rel(tdq.double().cpu().numpy(), odq)
    print("%-10s one step vs oracle: dq+ err max %.2e median %.2e  warnings %d" % (name, e.max(), np.median(e), bw.warnings()))
    B, T = 2048, 16
    tq, tdq = bw.to_device(q, dq, dtype)
    bw.step(tq, tdq, dt, T, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        tq, tdq = bw.to_device(q, dq, dtype)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        bw.step(tq, tdq, dt, T, **kw); torch.cuda.synchronize()
        t0 += time.perf_counter() - t1 - (time.perf_counter() - t1) + 0
    # timed region: the launches only
    ts = []
    for _ in range(reps):
        tq, tdq = bw.to_device(q, dq, dtype); torch.cuda.synchronize()
        a = time.perf_counter(); bw.step(tq, tdq, dt, T, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    print("%-10s %d worlds x %d steps: %.2f M world-steps/s   plan %s" % (name, B, T, B * T / min(ts) / 1e6,
          bw.plan(B, T, dtype=dtype, mixed=kw.get("mixed"))))
