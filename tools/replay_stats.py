#!/usr/bin/env python3
"""Error distribution of the float32 fused kernel against the float64 oracle over many (world, step) pairs of
the falling episode: every sampled state logged by the device is stepped once by the oracle, and every pair over the
1e-5 gate goes through tests/parity_tools.explain_outlier (a decision difference only counts when the decision was
marginal for the oracle itself) or the ill-conditioning rule of the tests.  Printed: the share within 1e-5 / 1e-4, the
outliers PER CRITERION (active, a-e, ill; parity_tools), the share of world-steps explained by (d) / (e), the largest
system errors (Y', v', the flipped solve's rows) among the cases that reached (d) / (e), the world-steps above the caps of
the tests (1e-3 in q, 1e-2 in dq) one by one, and the unexplained ones.
usage (GPU box): python tools/replay_stats.py [seed [world_stride [step_stride [contacts [kernels]]]]]   (defaults 1000, 16, 3, 4;
kernels: "body" = ARB_STEP_BODY_COLUMNS, "general" = ARB_STEP_GENERAL_KERNELS, default: what the library picks)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
from conftest import load_model
import parity_tools as P
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
m, _, _ = load_model("human36_c%d" % arg(4, 4))
bw = BatchedWorlds(m)
B, T, dt = 4096, 40, 5e-3
q, dq = synth.standing_states(m, B, seed=arg(1, 1000), drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
if len(sys.argv) > 5 and sys.argv[5] in ("body", "general"):
    P.KERNEL_KW.update({"body_columns" if sys.argv[5] == "body" else "general_kernels": True})
print("kernels:", P.KERNEL_KW or "default", bw.plan(B, T, other_inputs=True, **P.KERNEL_KW))
log = bw.rollout(tq, tdq, dt, T, cforce=bw.new_cforce(B, torch.float32), log_energy=False, **P.KERNEL_KW)
torch.cuda.synchronize()
worlds = np.arange(0, B, arg(2, 16))     # 256 worlds by default
errs, crit, unexplained, over, diag = [], {}, [], [], dict(e_adm=0., e_vel=0., e_row=0., apriori_adm=0., apriori_vel=0.)
for k in range(0, T - 1, arg(3, 3)):       # 13 steps by default
    eq, e = P.replay_errors(m, log["q"], log["dq"], (k,), worlds, dt)
    errs.append(np.maximum(e, eq))
    for i in np.flatnonzero((e > 1e-5) | (eq > 1e-5)):
        w = int(worlds[i])
        qf, dqf = log["q"][k][w].cpu().numpy(), log["dq"][k][w].cpu().numpy()
        why = P.explain_outlier(bw, m, qf, dqf, dt)
        if why is not None and why.criterion in ("d", "e"):
            for key in diag:
                diag[key] = max(diag[key], P.LAST_DIAG.get(key, 0.))
        if why is None:
            why = P.ill_conditioned(m, qf, dqf, dt, eq[i], e[i])
        tag = "UNEXPLAINED" if why is None else why.criterion
        crit[tag] = crit.get(tag, 0) + 1
        if why is None:
            unexplained.append((k, w, float(eq[i]), float(e[i]), dict(P.LAST_DIAG)))
        if not (eq[i] < 1e-3 and e[i] < 1e-2):
            over.append((k, w, float(eq[i]), float(e[i]), tag))
            print("  ABOVE THE CAP step %2d world %4d: err q %.2e dq %.2e [%s] -- %s" % (k, w, eq[i], e[i], tag, why))
    sys.stdout.flush()
e = np.concatenate(errs)
n = len(e)
print("all: %d pairs, within 1e-5: %.2f %%, within 1e-4: %.2f %%, max %.1e" % (n, 100 * (e <= 1e-5).mean(), 100 * (e <= 1e-4).mean(), e.max()))
print("outliers: %d, by criterion: %s" % (int((e > 1e-5).sum()), dict(sorted(crit.items()))))
de = crit.get("d", 0) + crit.get("e", 0)
print("criteria (d) + (e): %d = %.4f %% of the world-steps (cap %.2f %%); largest system errors among them: Y' %.1e, v' %.1e, solve rows %.1e"
      "; the a-priori float32 bounds they had to stay inside (parity_tools.APRIORI_K): Y' %.1e, v' %.1e of the largest entry"
      % (de, 100. * de / n, 100 * P.DE_SHARE_CAP, diag["e_adm"], diag["e_vel"], diag["e_row"], diag["apriori_adm"], diag["apriori_vel"]))
print("above the caps (q 1e-3, dq 1e-2): %d %s" % (len(over), over))
print("unexplained: %d %s" % (len(unexplained), unexplained[:20]))
