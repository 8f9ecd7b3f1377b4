#!/usr/bin/env python3
"""A/B of development builds of the library on the GPU box, in ONE process (round 5).

usage: tools/ab_r5.py [--contacts 4|8] [--batch 4096] [--rounds 3] [--dtype f32|f64] [--compare] name_a name_b ...
  name = a build under build/ab/<name>.so (tools/quick_build.sh), or "shipped" = arboris_python_amd/libarbstep.so
Times whole 40-step episodes of the headline workload per build, interleaved; prints M world-steps/s per build and round.
--compare: also the largest difference of the final states between the first build and each other one (0 = bit-identical),
and the share of worlds whose states differ.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from arboris_python_amd import scenes, synth, _capi            # noqa: E402
from arboris_python_amd.batch import BatchedWorlds             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--contacts", type=int, default=4)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--episodes", type=int, default=60)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--compare", action="store_true")
ap.add_argument("--general", action="store_true")
ap.add_argument("names", nargs="+")
a = ap.parse_args()

m = scenes.flat(scenes.human36_world(a.contacts))
dt_ = torch.float32 if a.dtype == "f32" else torch.float64
if a.contacts:
    q, dq = synth.world_states(m, range(a.batch), "standing", 1000, drop=0.03, vel=0.1)
else:
    q, dq = synth.world_states(m, range(a.batch), "random", 1000, angle=0.7, vel=1.0)
bws = {}
for n in a.names:
    path = _capi.LIB_PATH if n == "shipped" else os.path.join(ROOT, "build", "ab", n + ".so")
    bws[n] = BatchedWorlds(m, lib=_capi._open(path))
q0, dq0 = next(iter(bws.values())).to_device(q, dq, dt_)
kw = dict(general_kernels=True) if a.general else {}


def run(bw, n_ep):
    tq, tdq = q0.clone(), dq0.clone()
    cf = bw.new_cforce(a.batch, dt_) if m.nc else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_ep):
        tq.copy_(q0); tdq.copy_(dq0)
        if cf is not None:
            cf.zero_()
        bw.step(tq, tdq, 5e-3, a.steps, cforce=cf, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, tq, tdq


final = {}
for n, bw in bws.items():
    _, fq, fdq = run(bw, 3)
    final[n] = (fq.clone(), fdq.clone())
    print(n, "plan", bw.plan(a.batch, a.steps, dtype=dt_, **kw), "finite", bool(torch.isfinite(fq).all()), flush=True)
if a.compare:
    ref = a.names[0]
    for n in a.names[1:]:
        dqd = (final[n][1] - final[ref][1]).abs().max(dim=1).values / final[ref][1].abs().max(dim=1).values.clamp(min=1.)
        qd = (final[n][0] - final[ref][0]).abs().max(dim=1).values / final[ref][0].abs().max(dim=1).values.clamp(min=1.)
        d = torch.maximum(qd, dqd)
        print("compare %s vs %s: max rel diff %.3e, worlds differing %.4f, > 1e-5: %.4f, > 1e-3: %.4f"
              % (n, ref, float(d.max()), float((d > 0).double().mean()), float((d > 1e-5).double().mean()), float((d > 1e-3).double().mean())), flush=True)
for r in range(a.rounds):
    for n, bw in bws.items():
        el, _, _ = run(bw, a.episodes)
        print("round %d %s: %.3f M" % (r, n, a.batch * a.steps * a.episodes / el / 1e6), flush=True)
