#!/usr/bin/env python3
"""What would two worlds per wavefront in the Gauss-Seidel sweeps buy?  (round 3, before writing the kernel)

The device's own decision traces (arb_inspect_out.gs_trace: the decision of every executed local solve) of the bench
workload give, per world-step, the list of solves and their kinds.  With the measured cycle prices of DESIGN.md 3
(release / static solve ~0.9 k cycles, sliding solve ~3.4 k, everything outside the sweeps ~142 k per world-step) this
prints
  * the cost of the sweeps per world-step, alone and packed two worlds to a wavefront (a packed solve costs the MAX of
    the two worlds' solves, a sweep runs while either world still moves),
  * the episode cost per world and per pair and its spread (max / mean): with 4096 worlds on 2048 wave slots pairs
    cannot be rebalanced by the work queue -- the launch lasts as long as the slowest pair.
usage (GPU box): python tools/pack_model.py [contacts [batch [stride]]]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m, _, _ = load_model("human36_c%d" % nc)
bw = BatchedWorlds(m)
T, dt = 40, 5e-3
q, dq = synth.world_states(m, range(B), "standing", 1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
log = bw.rollout(tq, tdq, dt, T, cforce=cf, log_energy=False)
torch.cuda.synchronize()
C_FIXED, C_EASY, C_SLIDE, C_SWEEP = 142e3, 0.9e3, 3.4e3, 0.15e3
price = np.array([C_EASY, C_EASY, C_SLIDE, C_SLIDE + 6e3, C_EASY])          # codes 0 release, 1 static, 2 slide, 3 slide eig6, 4 other
alone = np.zeros((T, B)); packed = np.zeros((T, B // 2)); solves = np.zeros((T, 3))
for k in range(T):
    r = bw.inspect(log["q"][k], log["dq"][k], dt, ["gs_trace", "gs_stats"], cforce=bw.new_cforce(B, torch.float32))
    tr = r["gs_trace"].cpu().numpy()                      # (B, 20, nc), -1 = not executed
    st = r["gs_stats"].cpu().numpy()
    cost = np.where(tr >= 0, price[np.clip(tr, 0, 4)], 0.)  # (B, 20, nc)
    sweeps = (tr >= 0).any(axis=2)                        # (B, 20) sweep executed
    alone[k] = cost.sum(axis=(1, 2)) + C_SWEEP * sweeps.sum(axis=1)
    c2 = np.maximum(cost[0::2], cost[1::2])               # packed: a solve costs the dearer of the two
    s2 = sweeps[0::2] | sweeps[1::2]
    packed[k] = c2.sum(axis=(1, 2)) + C_SWEEP * s2.sum(axis=1)
    solves[k] = [(tr == 0).sum() + (tr == 1).sum(), (tr >= 2).sum(), sweeps.sum()]
    if k % 8 == 0 or k == T - 1:
        print("step %2d: sweeps cost per world alone %.0f k, packed %.0f k per pair (%.2f of two alone); easy %d sliding %d solves, %.1f sweeps"
              % (k, alone[k].mean() / 1e3, packed[k].mean() / 1e3, packed[k].mean() / (2 * alone[k].mean() + 1e-9),
                 solves[k][0], solves[k][1], solves[k][2] / B))
        sys.stdout.flush()
ga, gp = alone.mean(), packed.mean()
print("episode mean per world-step: sweeps alone %.1f k cycles, packed %.1f k per pair = %.1f k per world (efficiency %.2f; 0.5 = perfect)"
      % (ga / 1e3, gp / 1e3, gp / 2e3, gp / (2 * ga)))
tot_alone = (alone + C_FIXED).sum(axis=0)                 # per world, whole episode
tot_pair = (packed + 2 * C_FIXED).sum(axis=0)
print("episode cost per world: mean %.2f M cycles, max/mean %.3f, p99/mean %.3f" % (tot_alone.mean() / 1e6, tot_alone.max() / tot_alone.mean(), np.quantile(tot_alone, 0.99) / tot_alone.mean()))
print("episode cost per pair : mean %.2f M cycles, max/mean %.3f, p99/mean %.3f" % (tot_pair.mean() / 1e6, tot_pair.max() / tot_pair.mean(), np.quantile(tot_pair, 0.99) / tot_pair.mean()))
slots = 2048
# ideal makespans: unpacked with perfect balancing (the queue), packed with one pair per slot (B/2 <= slots) or balanced
un = tot_alone.sum() / slots
if B // 2 <= slots:
    pk = tot_pair.max()
else:
    pk = max(tot_pair.sum() / slots, tot_pair.max())
print("model makespan on %d slots: unpacked+queue %.2f M cycles, packed %.2f M cycles -> speed-up %.3f (perfectly balanced packed: %.3f)"
      % (slots, un / 1e6, pk / 1e6, un / pk, un / (tot_pair.sum() / min(slots, B // 2) if B // 2 <= slots else tot_pair.sum() / slots)))
