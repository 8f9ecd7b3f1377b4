#!/usr/bin/env python3
"""Time one arb_step launch (4096 worlds, human36 + 4 contacts, float32) in
different contact regimes, to see where the constraint stage spends its time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds

m = scenes.flat(scenes.human36_world(int(sys.argv[1]) if len(sys.argv) > 1 else 4))
bw = BatchedWorlds(m)
B = 4096


def timeit(q, dq, label, skip=False, reps=10):
    tq0, tdq0 = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    ts = []
    for r in range(reps + 2):
        tq, tdq = tq0.clone(), tdq0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bw.step(tq, tdq, 5e-3, 1, cforce=cf, skip_constraints=skip)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    r = bw.inspect(tq0, tdq0, 5e-3, ["c_active", "c_force"], skip_constraints=skip)
    act = r["c_active"].float().mean().item()
    print("%-34s %8.3f ms   active fraction %.2f  max|f| %.1f" % (label, 1e3 * np.median(ts[2:]), act, r["c_force"].abs().max().item()))


q, dq = synth.standing_states(m, B, seed=1, drop=0.0, vel=0.0)
qh = q.copy(); qh[:, 7] += 1.0
timeit(qh, dq, "skip_constraints flag", skip=True)
timeit(qh, dq, "1 m above ground (inactive)")
q2 = q.copy(); q2[:, 7] += 0.01
timeit(q2, dq, "1 cm above (active, release)")
timeit(q, dq, "resting on ground, zero vel")
q3 = q.copy(); q3[:, 7] -= 0.001
timeit(q3, dq, "1 mm penetration, zero vel")
dq4 = dq.copy(); dq4[:, 3] = 1.0
timeit(q3, dq4, "1 mm penetration, sliding 1 m/s")
q5, dq5 = synth.standing_states(m, B, seed=1, drop=0.03, vel=0.1)
timeit(q5, dq5, "bench initial states")
