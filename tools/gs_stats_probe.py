#!/usr/bin/env python3
"""Branch statistics of the Gauss-Seidel stage along the bench workload (GPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = scenes.flat(scenes.human36_world(nc))
bw = BatchedWorlds(m)
B = 4096
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
print("step  release  static  slide_fast  slide_eig6  sweeps  (mean per world; max slide_eig6 in a world)")
for k in range(40):
    if k % 4 == 0 or k > 34:
        r = bw.inspect(tq, tdq, 5e-3, ["gs_stats", "stamps"])
        st = r["gs_stats"].float()
        ph = (r["stamps"][:, 1:] - r["stamps"][:, :-1]).double().mean(0).tolist()
        print("%3d  %7.1f %7.1f %10.1f %10.2f %6.1f   max %d   cycles A %.0f A' %.0f B %.0f C %.0f D %.0f GS %.0f E %.0f"
              % ((k,) + tuple(st.mean(0).tolist()) + (int(st[:, 3].max()),) + tuple(ph)))
        print("      sweeps histogram:", torch.bincount(r["gs_stats"][:, 4].long(), minlength=21).tolist())
        tot = (r["stamps"][:, -1] - r["stamps"][:, 0]).double()
        gs = (r["stamps"][:, 6] - r["stamps"][:, 5]).double()
        qs = torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64, device=tot.device)
        print("      per-world total cycles p50/p90/p99/max: %s   GS: %s" % (
            " ".join("%.0fk" % (v / 1e3) for v in torch.quantile(tot, qs).tolist()),
            " ".join("%.0fk" % (v / 1e3) for v in torch.quantile(gs, qs).tolist())))
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
print("finite", bool(torch.isfinite(tq).all()), "max |dq|", float(tdq.abs().max()))
