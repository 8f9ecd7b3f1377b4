#!/usr/bin/env python3
"""Soak: many seeds of the falling-human workload at full size -- finiteness of every world, fused == wave-split
(general kernels, bitwise) on a subsample, rollout == plain step (bitwise) on the whole batch, and the fraction of sampled world-steps within 1e-5 of the oracle.
usage (GPU box): python tools/soak.py [nseeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import arb_oracle as O
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for nc in (4, 8):
    m = scenes.flat(scenes.human36_world(nc))
    bw = BatchedWorlds(m)
    B, T, dt = 65536 if nc == 4 else 16384, 40, 5e-3
    for seed in range(nseeds):
        q, dq = synth.standing_states(m, B, seed=7000 + seed, drop=0.03, vel=0.3 + 0.1 * seed)
        q[:, 7] -= 0.005 * seed
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        t0 = time.perf_counter()
        log = bw.rollout(tq, tdq, dt, T, cforce=cf, log_energy=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        fin = bool(torch.isfinite(tq).all() and torch.isfinite(tdq).all() and torch.isfinite(cf).all())
        # a subsample through the general kernels, fused and wave-split (the split execution runs the general kernels; the
        # eight-contact model's default are the body-space-column kernels, equal to those to rounding only): bit patterns
        sub = np.arange(seed, B, 8)[:4096]
        bits = lambda t: t.contiguous().view(torch.int32)            # (bit patterns: NaN == NaN)
        res = []
        for split in (False, "wave"):
            sq, sdq = bw.to_device(q[sub], dq[sub], torch.float32)
            scf = bw.new_cforce(len(sub), torch.float32)
            bw.step(sq, sdq, dt, T, cforce=scf, split=split, general_kernels=not split)
            torch.cuda.synchronize()
            res.append((sq, sdq, scf))
        same = all(bool(torch.equal(bits(a_), bits(b_))) for a_, b_ in zip(*res))
        # the whole batch again through the plain step (no logs: the FEAT 0 kernel with the work queue; the rollout above
        # is the FEAT 3 kernel) -- bit patterns must be equal
        pq, pdq = bw.to_device(q, dq, torch.float32)
        pcf = bw.new_cforce(B, torch.float32)
        bw.step(pq, pdq, dt, T, cforce=pcf)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(bits(pq), bits(tq)) and torch.equal(bits(pdq), bits(tdq)) and torch.equal(bits(pcf), bits(cf)))
        ws = np.arange(3 + seed, B, B // 24)[:24]
        ok = []
        for k in (8, 20, 33):
            oq, odq, _ = O.step(m, log["q"][k][ws].double().cpu().numpy(), log["dq"][k][ws].double().cpu().numpy(), dt)
            e = np.maximum(np.abs(log["q"][k + 1][ws].cpu().numpy() - oq).max(1) / np.maximum(1, np.abs(oq).max(1)),
                           np.abs(log["dq"][k + 1][ws].cpu().numpy() - odq).max(1) / np.maximum(1, np.abs(odq).max(1)))
            ok.append(e < 1e-5)
        ok = np.concatenate(ok)
        print("nc %d seed %d: %.1f M world-steps/s (with logs), finite %s, wave-split and plain-step bitwise %s, within 1e-5: %d/%d, max |dq| %.0f, max force %.0f"
              % (nc, seed, B * T / el / 1e6, fin, same, ok.sum(), len(ok), float(tdq.abs().max()), float(cf.abs().max())), flush=True)
        if not fin:
            # an exploding world is acceptable only if the float64 reference algorithm explodes on it too
            bad = torch.nonzero(~(torch.isfinite(tdq).all(dim=1) & torch.isfinite(tq).all(dim=1)))[:, 0].cpu().numpy()
            big = torch.nonzero(tdq.abs().amax(dim=1) > 1e4)[:, 0].cpu().numpy()
            print("   non-finite worlds: %d, |dq| > 1e4: %d of %d" % (len(bad), len(big), B))
            for wbad in bad[:3]:
                oq, odq, ocf = q[wbad:wbad + 1].astype(np.float32).astype(np.float64), dq[wbad:wbad + 1].astype(np.float32).astype(np.float64), None
                hist = []
                for k in range(T):
                    oq, odq, ocf = O.step(m, oq, odq, dt, ocf)
                    hist.append(float(np.abs(odq).max()))
                dev = [float(log["dq"][k][wbad].abs().max()) for k in range(T)]
                print("   world %d: oracle max|dq| per step %s" % (wbad, " ".join("%.0e" % x for x in hist[::3])))
                print("            device max|dq| per step %s" % " ".join("%.0e" % x for x in dev[::3]))
                assert not np.isfinite(hist).all() or max(hist) > 1e3, "device overflowed on a world the oracle keeps bounded"
        assert same
    bw.close()
print("soak ok")
