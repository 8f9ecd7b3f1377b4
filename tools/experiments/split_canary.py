"""The one input (human36 + 4 contacts, world 31974 of the 65536-world test at step 19) on which a build of the
lane-per-world Gauss-Seidel kernel returned a 1e18 N contact force (split execution): steps it with
the split and the fused execution of the library in ARBSTEP_LIB and prints both against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch
import arb_oracle as O
from conftest import load_model
from arboris_python_amd.batch import BatchedWorlds
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
d = np.load(os.path.join(ROOT, "tests", "golden", "canary_eig6_fallback.npz"))
dt = 5e-3
oq, odq, ocf = O.step(m, d["q"].astype(np.float64), d["dq"].astype(np.float64), dt)
for mode in ("split", "fused"):
    a = torch.as_tensor(d["q"], dtype=torch.float32, device=bw.device); b = torch.as_tensor(d["dq"], dtype=torch.float32, device=bw.device)
    c = bw.new_cforce(1, torch.float32)
    bw.step(a, b, dt, 1, cforce=c, fused=(mode == "fused"), split=(mode == "split"))
    torch.cuda.synchronize()
    print(os.environ.get("ARBSTEP_LIB", "default"), mode, "max|dq| %.4g  cf %s   oracle cf %s" % (float(b.abs().max()), c.cpu().numpy().round(3)[0, 2].tolist(), np.round(ocf, 3)[0, 2].tolist()))
