#!/usr/bin/env python3
"""Where does a float32 step lose accuracy?  Stage-by-stage relative errors (device float32 inspect outputs against
the float64 oracle on the same float32-rounded inputs) for one randomised model of tests/test_gpu_random_models.py.
usage (GPU box): python tools/f32_stage_errors.py [seed | rand4 | rand8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import arb_oracle as O
from test_gpu_random_models import random_world
from arboris_python_amd.flatten import flatten_world
from arboris_python_amd.batch import BatchedWorlds
from arboris_python_amd import synth
if len(sys.argv) > 1 and sys.argv[1].startswith("rand"):
    # the hardest golden states of the headline model: tests/golden/g3_contacts.npz rand4 / rand8
    from conftest import load_golden, load_model
    nc = int(sys.argv[1][4:])
    m, _, _ = load_model("human36_c%d" % nc)
    g = load_golden("g3_contacts.npz")
    q, dq, dt = g["rand%d_q" % nc], g["rand%d_dq" % nc], 5e-3
    B = len(q)
else:
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    w = random_world(seed)
    m, q0, dq0 = flatten_world(w)
    B = 12
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1))
    qr, dqr = synth.random_states(m, B, seed=seed, angle=0.8, vel=1.5, root_box=((-.3, .3), (-.2, .5), (-.3, .3)))
    q[1:], dq[1:] = qr[1:], dqr[1:]
    dt = float(np.random.default_rng(100 + seed).choice([2e-3, 5e-3]))
f = lambda a: np.asarray(a, np.float32).astype(np.float64)
q, dq = f(q), f(dq)
print("model: jtype", list(map(int, m.jtype)), "ctype", list(map(int, m.ctype)), "ndof", m.ndof, "dt", dt)
oq, odq, ocf, d = O.step(m, q, dq, dt, cforce=np.zeros((B, m.nc, 4)), debug=True)
bw = BatchedWorlds(m)
rel = lambda a, b: np.abs(np.asarray(a, np.float64) - b).reshape(B, -1).max(1) / np.maximum(1e-30, np.abs(b).reshape(B, -1).max(1))
for dtype in (torch.float64, torch.float32):
    tq, tdq = bw.to_device(q, dq, dtype)
    r = bw.inspect(tq, tdq, dt, ["pose", "twist", "M", "N", "Z", "gforce0", "vel_free", "c_sdist", "c_active", "c_jac", "c_force", "dq_next", "q_next"],
                   cforce=bw.new_cforce(B, dtype))
    r = {k: v.cpu().numpy() for k, v in r.items()}
    print(str(dtype))
    rhs = (d["M"] @ (dq / dt)[..., None])[..., 0] + d["gforce0"]
    vfree = np.linalg.solve(d["Z"], rhs[..., None])[..., 0]
    for name, got, ref in (("pose", r["pose"], d["pose"]), ("twist", r["twist"], d["twist"]), ("M", r["M"], d["M"]), ("N", r["N"], d["N"]),
                           ("Z", r["Z"], d["Z"]), ("gforce0", r["gforce0"], d["gforce0"]), ("vel_free", r["vel_free"], vfree),
                           ("c_sdist", r["c_sdist"], d["sdist"]), ("c_jac", r["c_jac"], d["jac"].reshape(B, m.nc, 4, m.ndof)),
                           ("c_force", r["c_force"], ocf), ("dq_next", r["dq_next"], odq), ("q_next", r["q_next"], oq)):
        e = rel(got, ref)
        print("  %-9s max rel err %.2e   (per world: %s)" % (name, e.max(), " ".join("%.0e" % x for x in e)))
    print("  active equal:", bool((r["c_active"].astype(bool) == d["active"]).all()), " |dq+| per world:", " ".join("%.0f" % x for x in np.abs(odq).max(1)),
          " cond(Z):", " ".join("%.0e" % np.linalg.cond(z) for z in d["Z"][:4]))
