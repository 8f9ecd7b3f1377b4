#!/usr/bin/env python3
"""Fused step kernel against the split executions (Gauss-Seidel sweeps in their own kernel: one lane per world,
or one wavefront per world) over batch sizes; whole falling episodes, human36 + 4 or 8 contacts, float32.
usage (GPU box): python tools/split_sweep.py [--contacts 4] [--batches 2048,4096,8192,16384,65536]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds

ap = argparse.ArgumentParser()
ap.add_argument("--contacts", type=int, default=4)
ap.add_argument("--batches", default="2048,4096,8192,16384,65536")
ap.add_argument("--modes", default="fused,lane,wave")
ap.add_argument("--episodes", type=int, default=6)
ap.add_argument("--spl", type=int, default=40, help="steps per arb_step call")
a = ap.parse_args()
m = scenes.flat(scenes.human36_world(a.contacts))
bw = BatchedWorlds(m, 0)
EP, dt = 40, 5e-3
out = {}
for B in [int(x) for x in a.batches.split(",")]:
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    q0, dq0 = bw.to_device(q, dq, torch.float32)
    row = {}
    for mode in a.modes.split(","):
        split = {"fused": False, "lane": True, "wave": "wave", "mfma": False}[mode]
        tq, tdq, cf = q0.clone(), dq0.clone(), bw.new_cforce(B, torch.float32)
        def episode():
            tq.copy_(q0); tdq.copy_(dq0)
            k = 0
            while k < EP:
                c = min(a.spl, EP - k)
                bw.step(tq, tdq, dt, c, cforce=cf, split=split, mfma=(mode == "mfma"))
                k += c
        episode(); torch.cuda.synchronize()
        n = max(2, min(a.episodes, int(a.episodes * 8192 / B) + 1))
        t0 = time.perf_counter()
        for _ in range(n):
            episode()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        row[mode] = B * EP * n / el
        row[mode + "_finite"] = bool(torch.isfinite(tq).all())
    out[B] = row
    print(B, {k: (round(v / 1e6, 2) if not isinstance(v, bool) else v) for k, v in row.items()}, flush=True)
print(json.dumps({"contacts": a.contacts, "spl": a.spl, "world_steps_per_s": out}))
