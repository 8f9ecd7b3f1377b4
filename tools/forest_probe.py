#!/usr/bin/env python3
"""How much does it pay to put several small worlds into one wavefront (a forest of K copies of the model,
flatten.replicate_model)?  For each model and K: world-steps/s of a multi-step launch, and the largest difference
of the final state from the K = 1 launch.  usage (GPU box): python tools/forest_probe.py [worlds] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_model
from arboris_python_amd.batch import BatchedWorlds
from arboris_python_amd.flatten import replicate_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40320          # divisible by 1..10, 12, 14, 15, 16, 18, 20, 21
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for name, Ks, dts in (("simplearm", (1, 2, 3, 4, 5, 7, 10, 15, 21), ("f32", "f64")),
                      ("snake9_free_g", (1, 2, 3, 4, 5, 7), ("f32",)),
                      ("jointlimits_min", (1, 2, 3, 4, 5, 7, 10), ("f32",)),
                      ("ballsocket", (1, 2, 3, 4, 5, 7), ("f32",)),
                      ("planar_contact_slide", (1, 2, 3, 4, 5), ("f32",))):
    m, q0, dq0 = load_model(name)
    rng = np.random.default_rng(3)
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + 0.2 * rng.standard_normal((B, m.ndof))
    lin = m.dof2q >= 0
    q[:, m.dof2q[lin]] += 0.2 * rng.standard_normal((B, int(lin.sum())))
    for dn in dts:
        dtype = torch.float32 if dn == "f32" else torch.float64
        ref = None
        for K in Ks:
            if B % K:
                continue
            f = replicate_model(m, K)
            try:
                bw = BatchedWorlds(f)
            except Exception as e:
                print("%-22s %s K=%2d: %s" % (name, dn, K, e)); continue
            tq, tdq = bw.to_device(q.reshape(B // K, -1), dq.reshape(B // K, -1), dtype)
            cf = bw.new_cforce(B // K, dtype) if f.nc else None
            a, b = tq.clone(), tdq.clone()
            bw.step(a, b, 1e-3, T, cforce=cf)
            torch.cuda.synchronize()
            res = (a.reshape(B, -1).cpu().numpy(), b.reshape(B, -1).cpu().numpy())
            if ref is None:
                ref = res
            err = max(np.abs(res[0] - ref[0]).max(), np.abs(res[1] - ref[1]).max())
            ts = []
            for _ in range(5):
                a.copy_(tq); b.copy_(tdq)
                if cf is not None:
                    cf.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                bw.step(a, b, 1e-3, T, cforce=cf)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print("%-22s %s K=%2d (ndof %2d, tile %2d, %d sets): %8.2f M world-steps/s   max |diff| vs K=1 %.2e"
                  % (name, dn, K, f.ndof, bw.info["nmax"], bw.info["nsets"], B * T / min(ts) / 1e6, err), flush=True)
            bw.close()
