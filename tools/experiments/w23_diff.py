#!/usr/bin/env python3
"""Development: where do the two-wave and three-wave builds of the float32 step kernel part?  Steps the same batch one
launch per step with ARB_FORCE_WAVES=2 and =3 and reports the first (step, world) whose states differ in any bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
B, T, dt = 5000, 40, 5e-3
q, dq = synth.world_states(m, range(B), "standing", 77, drop=0.03, vel=0.1)
st = {}
for wv in ("2", "3"):
    os.environ["ARB_FORCE_WAVES"] = wv
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    hist = []
    for k in range(T):
        hist.append((tq.clone(), tdq.clone(), cf.clone()))
        bw.step(tq, tdq, dt, 1, cforce=cf)
    torch.cuda.synchronize()
    hist.append((tq.clone(), tdq.clone(), cf.clone()))
    st[wv] = hist
for k in range(T + 1):
    a, b = st["2"][k], st["3"][k]
    d = (a[0] != b[0]).any(dim=1) | (a[1] != b[1]).any(dim=1)
    if d.any():
        ws = torch.nonzero(d).flatten().cpu().numpy()
        print("first difference after step", k - 1, "worlds", ws[:10], "count", len(ws))
        w = int(ws[0])
        pq, pdq, pcf = st["2"][k - 1]
        print("  same inputs:", bool((st["2"][k - 1][0][w] == st["3"][k - 1][0][w]).all()), bool((st["2"][k - 1][2][w] == st["3"][k - 1][2][w]).all()))
        print("  max |dq2 - dq3|", float((a[1][w] - b[1][w]).abs().max()), "max|dq|", float(a[1][w].abs().max()))
        print("  cforce 2:", a[2][w].cpu().numpy().round(3).tolist()); print("  cforce 3:", b[2][w].cpu().numpy().round(3).tolist())
        r = bw.inspect(pq[w:w + 1].contiguous(), pdq[w:w + 1].contiguous(), dt, ["gs_stats", "c_active", "c_sdist"], cforce=pcf[w:w + 1].contiguous())
        print("  gs_stats (release, static, slide, slide eig6, sweeps):", r["gs_stats"].cpu().numpy()[0], "active", r["c_active"].cpu().numpy()[0])
        np.savez(ROOT + "/gpurun_out/w23_case.npz", q=pq[w].cpu().numpy(), dq=pdq[w].cpu().numpy(), cf=pcf[w].cpu().numpy())
        break
else:
    print("no difference over", T, "steps")
