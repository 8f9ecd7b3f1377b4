#!/usr/bin/env python3
"""Throughput of the wide kernels (worlds past one wavefront).  usage (GPU box): python tools/wide_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.flatten import flatten_world
from arboris_python_amd.batch import BatchedWorlds

def leg(name, m, q, dq, dt, T, dtype, cf):
    bw = BatchedWorlds(m)
    B = len(q)
    ts = []
    for _ in range(4):
        tq, tdq = bw.to_device(q, dq, dtype)
        c = bw.new_cforce(B, dtype) if cf else None
        torch.cuda.synchronize(); a = time.perf_counter()
        bw.step(tq, tdq, dt, T, cforce=c); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    print("%-28s %5d worlds x %3d steps %s: %.3f M world-steps/s (%.2f ms/launch)  info %s" % (name, B, T, str(dtype)[6:], B * T / min(ts) / 1e6, min(ts) * 1e3,
          {k: bw.info[k] for k in ("ndof", "nc", "wide", "lds_bytes_f64")}), "finite", bool(torch.isfinite(tdq).all()))
    bw.close()

for nl in (64, 100, 128, 256):
    m = scenes.flat(scenes.snake_world(nl))
    for B in (256, 2048):
        q, dq = synth.random_states(m, B, seed=0, angle=0.5, vel=1.0)
        leg("snake-%d" % nl, m, q, dq, 1e-3, 16, torch.float64, False)
for nobj in (4, 12):                     # (66 dofs / 8 contacts: 99 columns; 114 dofs / 16 contacts: 179 columns)
    w = scenes.human36_and_objects_world(nobj)
    m, q0, dq0 = flatten_world(w)
    for B in (512, 4096):
        rng = np.random.default_rng(0)
        q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + rng.uniform(-0.1, 0.1, (B, m.ndof))
        for dtype in (torch.float32, torch.float64):
            leg("human36 + %d objects" % nobj, m, q, dq, 5e-3, 40, dtype, True)
for nballs in (3, 8):                    # (every pair of get_all_contacts: 38 contacts on 60 dofs, 108 on 90; 10-30 active)
    m, q0, dq0 = flatten_world(scenes.human36_and_balls_world(nballs))
    B = 512
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + np.random.default_rng(0).uniform(-0.05, 0.05, (B, m.ndof))
    leg("human36 + %d balls, all pairs" % nballs, m, q, dq, 5e-3, 40, torch.float64, True)
