import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import load_golden, load_model
from arboris_python_amd.batch import BatchedWorlds
import arb_oracle as O
g3 = load_golden("g3_contacts.npz")
for nc in (4, 8):
    m, _, _ = load_model("human36_c%d" % nc)
    bw = BatchedWorlds(m)
    Q, DQ = g3["drop%d_q" % nc][:39], g3["drop%d_dq" % nc][:39]
    Q32, DQ32 = Q.astype(np.float32).astype(np.float64), DQ.astype(np.float32).astype(np.float64)
    oq, odq, ocf, d = O.step(m, Q32, DQ32, 5e-3, debug=True)        # oracle on the f32-representable inputs
    tq, tdq = bw.to_device(Q, DQ, torch.float32)
    cf = bw.new_cforce(39, torch.float32)
    bw.step(tq, tdq, 5e-3, 1, cforce=cf)
    torch.cuda.synchronize()
    gq, gdq = tq.cpu().numpy().astype(np.float64), tdq.cpu().numpy().astype(np.float64)
    e_ref = np.abs(gdq - g3["drop%d_dq" % nc][1:]).max(1)
    e_or = np.abs(gdq - odq).max(1)
    print("nc=%d  step: err vs reference(f64 inputs) | err vs oracle(f32-rounded inputs) | active" % nc)
    for k in range(39):
        if e_ref[k] > 3e-6 or k % 8 == 0:
            print("  %2d  %.2e  %.2e  %s" % (k, e_ref[k], e_or[k], d["active"][k].astype(int)))
    print("  max: %.2e  %.2e   (max|dq| %.2f)" % (e_ref.max(), e_or.max(), np.abs(odq).max()))
    bw.close()
