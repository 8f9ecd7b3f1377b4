#!/usr/bin/env python3
"""Fused kernel against the lane-per-world (split) execution over a whole 65536-world batch, at several points
of the falling episode; worlds on which they disagree are replayed through the float64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
import arb_oracle as O
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
name = sys.argv[1] if len(sys.argv) > 1 else "human36_c4"
m, _, _ = load_model(name)
bw = BatchedWorlds(m)
B, dt = 65536, 5e-3
q, dq = synth.standing_states(m, B, seed=11, drop=0.03, vel=0.1)
q[:, 7] -= 0.02
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
def werr(a, b):
    a = a.double(); b = b.double()
    return ((a - b).abs().amax(dim=1) / torch.clamp(b.abs().amax(dim=1), min=1.)).cpu().numpy()
for k in range(36):
    if k % 5 == 0:
        fa, fb = tq.clone(), tdq.clone(); fc = bw.new_cforce(B, torch.float32)
        sa, sb = tq.clone(), tdq.clone(); sc = bw.new_cforce(B, torch.float32)
        bw.step(fa, fb, dt, 1, cforce=fc, fused=True)
        bw.step(sa, sb, dt, 1, cforce=sc, split="wave")
        torch.cuda.synchronize()
        e = np.maximum(werr(fa, sa), werr(fb, sb))
        bad = np.nonzero(~(e < 1e-4))[0]
        line = "step %2d: fused vs split  median %.1e  p99.9 %.1e  max %.1e  worlds > 1e-4: %d" % (k, np.median(e), np.quantile(e, 0.999), np.nanmax(e), len(bad))
        if len(bad):
            w = bad[:8]
            oq, odq, _ = O.step(m, tq[w].double().cpu().numpy(), tdq[w].double().cpu().numpy(), dt)
            ef = np.abs(fb[w].double().cpu().numpy() - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
            es = np.abs(sb[w].double().cpu().numpy() - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
            line += "   vs oracle dq: fused %s  split %s" % (np.array2string(ef, precision=1), np.array2string(es, precision=1))
            qf = np.abs(fa[w].double().cpu().numpy() - oq).max(axis=1); qs = np.abs(sa[w].double().cpu().numpy() - oq).max(axis=1)
            line += "  q: fused %s split %s (worst q index fused %s split %s)" % (
                np.array2string(qf, precision=1), np.array2string(qs, precision=1),
                np.abs(fa[w].double().cpu().numpy() - oq).argmax(axis=1).tolist(), np.abs(sa[w].double().cpu().numpy() - oq).argmax(axis=1).tolist())
        print(line); sys.stdout.flush()
    bw.step(tq, tdq, dt, 1, cforce=cf)
