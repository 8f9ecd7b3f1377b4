#!/usr/bin/env python3
"""Anatomy of one float32 outlier: python tools/outlier_probe.py seed step world [step world ...]   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
import arb_oracle as O
from conftest import load_model, oracle_sensitivity
from parity_tools import explain_outlier, solve_margins
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
NCON = int(os.environ.get("CONTACTS", "4"))
m, _, _ = load_model("human36_c%d" % NCON)
bw = BatchedWorlds(m)
B, T, dt = 4096, 40, 5e-3
seed = int(sys.argv[1])
q, dq = synth.standing_states(m, B, seed=seed, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), log_energy=False)
torch.cuda.synchronize()
args = [int(x) for x in sys.argv[2:]]
for k, w in zip(args[0::2], args[1::2]):
    qf, dqf = log["q"][k][w].cpu().numpy(), log["dq"][k][w].cpu().numpy()
    tr = []
    oq, odq, ocf, d = O.step(m, qf[None].astype(np.float64), dqf[None].astype(np.float64), dt, debug=True, trace=tr)
    g = log["dq"][k + 1][w].double().cpu().numpy()
    e = np.abs(g - odq[0]); i = int(e.argmax())
    print("== seed %d step %d world %d: max |ddq| %.3e at dof %d (oracle %.6f device %.6f), |dq|max %.3f" % (seed, k, w, e.max(), i, odq[0][i], g[i], np.abs(odq).max()))
    print("   explain:", explain_outlier(bw, m, qf, dqf, dt))
    r = bw.inspect(log["q"][k][w:w + 1].contiguous(), log["dq"][k][w:w + 1].contiguous(), dt,
                   ["gs_stats", "gs_trace", "c_active", "c_force", "c_sdist", "vel_free", "dq_next", "c_adm", "c_vel"], cforce=bw.new_cforce(1, torch.float32))
    print("   device gs_stats", r["gs_stats"].cpu().numpy()[0], "active", r["c_active"].cpu().numpy()[0], "oracle active", d["active"][0].astype(int))
    print("   sdist dev", r["c_sdist"].cpu().numpy()[0], "oracle", d["sdist"][0])
    cfd, cfo = r["c_force"].cpu().numpy()[0], ocf[0]
    print("   force dev\n", cfd.round(4), "\n   force oracle\n", cfo.round(4), "\n   max |df| %.3e rel %.2e" % (np.abs(cfd - cfo).max(), np.abs(cfd - cfo).max() / max(1., np.abs(cfo).max())))
    # velocity without constraint forces
    Y = d["Y"][0]; rhs = d["M"][0] @ (dqf.astype(np.float64) / dt) + d["gforce0"][0]
    vfree = Y @ rhs
    print("   vel_free err %.2e" % (np.abs(r["vel_free"].cpu().numpy()[0] - vfree).max() / max(1., np.abs(vfree).max())))
    dtr = r["gs_trace"].cpu().numpy()[0]
    otr = -np.ones_like(dtr)
    for t in tr: otr[t["sweep"], t["c"]] = t["branch"]
    nsw = int(r["gs_stats"].cpu().numpy()[0][4])
    dd = [(s, c, int(otr[s, c]), int(min(dtr[s, c], 2))) for s in range(nsw) for c in range(m.nc) if otr[s, c] >= 0 and min(dtr[s, c], 2) != otr[s, c]]
    print("   decision differences (sweep, contact, oracle, device):", dd[:8], "device sweeps", nsw)
    mg = [(t["sweep"], t["c"]) + tuple(None if x is None else float("%.2e" % x) for x in solve_margins(t)) for t in tr]
    small = sorted(mg, key=lambda x: min(v for v in x[2:] if v is not None))[:4]
    print("   smallest oracle margins (sweep, c, release, cone):", small)
    e_adm = np.abs(r["c_adm"].double().cpu().numpy()[0] - d["adm"][0]).max() / np.abs(d["adm"][0]).max()
    e_vel = np.abs(r["c_vel"].double().cpu().numpy()[0] - d["vel0"][0]).max() / max(1., np.abs(d["vel0"][0]).max())
    print("   device system vs oracle: Y' %.2e  v' %.2e" % (e_adm, e_vel))
    from parity_tools import sweeps_on
    got = sweeps_on(m, r["c_adm"].double().cpu().numpy()[0], r["c_vel"].double().cpu().numpy()[0], r["c_sdist"].double().cpu().numpy()[0], r["c_active"].cpu().numpy()[0].astype(bool), dt)
    dd2 = [(s, c, int(got[s, c]), int(min(dtr[s, c], 2))) for s in range(nsw) for c in range(m.nc) if got[s, c] >= 0 and min(dtr[s, c], 2) != got[s, c]]
    print("   float64 sweeps on the device system vs device trace, differences (sweep, c, f64, device):", dd2[:8])
    sq, sdq = oracle_sensitivity(m, qf[None], dqf[None], dt, samples=8)
    print("   oracle sensitivity to one float32 ulp on the input: q %.2e dq %.2e" % (sq[0], sdq[0]))
    # condition of the problem
    print("   cond(Z) %.2e  cond(adm) %.2e" % (np.linalg.cond(d["Z"][0]), np.linalg.cond(d["adm"][0][np.ix_(d["active"][0].repeat(4), d["active"][0].repeat(4))]) if d["active"][0].any() else 0.))
