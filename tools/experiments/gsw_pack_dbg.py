import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m)
B = 3
q, dq = synth.world_states(m, range(B), "standing", 11, drop=0.03, vel=0.2)
q[:, 7] -= 0.035
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for mode in ("wave", "pack"):
    os.environ["ARB_GSW_PACK"] = "1" if mode == "pack" else "0"
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    bw.step(tq, tdq, 5e-3, 1, cforce=cf, split="wave", waves=2)
    torch.cuda.synchronize()
    print(mode); print(cf.cpu().numpy().reshape(B, -1))
