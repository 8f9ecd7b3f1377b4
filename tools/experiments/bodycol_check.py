#!/usr/bin/env python3
"""Body-space constraint columns (FEAT bit 16) against the oracle and against the two-column-set kernels (development).
usage: bodycol_check.py <build under build/ab | shipped>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import arb_oracle as O
from arboris_python_amd import scenes, synth, _capi
from arboris_python_amd.batch import BatchedWorlds
name = sys.argv[1]
path = _capi.LIB_PATH if name == "shipped" else os.path.join(ROOT, "build", "ab", name + ".so")
m = scenes.flat(scenes.human36_world(8))
bw = BatchedWorlds(m, lib=_capi._open(path))
rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b).max(axis=-1) / np.maximum(1., np.abs(b).max(axis=-1))))
B = 256
q, dq = synth.standing_states(m, B, seed=5, drop=0.03, vel=0.1)
q[:, 7] -= 0.012
for dtype in (torch.float64, torch.float32):
    print(dtype, "plan", bw.plan(B, 1, dtype=dtype), "general", bw.plan(B, 1, dtype=dtype, general_kernels=True)["feat"])
    npt = np.float64 if dtype == torch.float64 else np.float32
    qi, dqi = q.astype(npt).astype(np.float64), dq.astype(npt).astype(np.float64)
    oq, odq, ocf, dbg = O.step(m, qi, dqi, 5e-3, debug=True)
    for gk in (False, True):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype)
        try:
            bw.step(tq, tdq, 5e-3, 1, cforce=cf, general_kernels=gk)
        except Exception as e:
            print("  general" if gk else "  bodycol", "step failed:", e); continue
        torch.cuda.synchronize()
        print("  %s: q %.2e dq %.2e cforce %.2e (max force %.0f)" % ("general" if gk else "bodycol", rel(tq.cpu().numpy(), oq), rel(tdq.cpu().numpy(), odq),
              np.abs(cf.cpu().numpy().reshape(B, -1) - ocf.reshape(B, -1)).max() / np.abs(ocf).max(), np.abs(ocf).max()))
    if dtype == torch.float32:
        tq, tdq = bw.to_device(q, dq, dtype)
        r = bw.inspect(tq, tdq, 5e-3, ["c_adm", "c_vel", "c_jac", "c_active", "c_force", "dq_next", "gforce"], cforce=bw.new_cforce(B, dtype))
        torch.cuda.synchronize()
        print("  inspect: dq_next %.2e c_force %.2e" % (rel(r["dq_next"].cpu().numpy(), odq), np.abs(r["c_force"].cpu().numpy().reshape(B, -1) - ocf.reshape(B, -1)).max() / np.abs(ocf).max()))
        for k, ok in (("c_adm", "adm"), ("c_vel", "vel0"), ("c_jac", "cjac")):
            if ok in dbg:
                a = r[k].double().cpu().numpy().reshape(B, -1); b = np.asarray(dbg[ok]).reshape(B, -1)
                print("  inspect %s vs oracle %s: %.2e of max" % (k, ok, np.abs(a - b).max() / np.abs(b).max()))
        print("  oracle debug keys:", sorted(dbg.keys()))
# episodes: body-space against the two-set kernels, and the rates
B = 4096
q, dq = synth.world_states(m, range(B), "standing", 1000, drop=0.03, vel=0.1)
for dtype in (torch.float64, torch.float32):
    fin = {}
    for gk in (False, True):
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype)
        try:
            bw.step(tq, tdq, 5e-3, 12, cforce=cf, general_kernels=gk)
        except Exception as e:
            print("episode", dtype, gk, "failed", e); continue
        torch.cuda.synchronize()
        bw.status()
        fin[gk] = (tq, tdq)
    if len(fin) == 2:
        d = ((fin[True][1] - fin[False][1]).abs().max(dim=1).values / fin[True][1].abs().max(dim=1).values.clamp(min=1.))
        print("12 steps %s: bodycol vs general dq: max %.2e, share > 1e-5: %.4f" % (dtype, float(d.max()), float((d > 1e-5).double().mean())))
q0, dq0 = bw.to_device(q, dq, torch.float32)
for gk in (False, True, False, True):
    tq, tdq = q0.clone(), dq0.clone(); cf = bw.new_cforce(B, torch.float32)
    try:
        bw.step(tq, tdq, 5e-3, 40, cforce=cf, general_kernels=gk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            tq.copy_(q0); tdq.copy_(dq0); cf.zero_()
            bw.step(tq, tdq, 5e-3, 40, cforce=cf, general_kernels=gk)
        torch.cuda.synchronize()
        print("rate %s: %.2f M  plan %s" % ("general" if gk else "bodycol", B * 40 * 30 / (time.perf_counter() - t0) / 1e6, bw.plan(B, 40, general_kernels=gk)))
    except Exception as e:
        print("rate", gk, "failed", e)
