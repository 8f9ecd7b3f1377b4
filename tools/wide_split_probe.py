import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/arboris_python_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from arboris_python_amd import scenes
from arboris_python_amd.flatten import flatten_world
from arboris_python_amd.batch import BatchedWorlds
w = scenes.human36_and_objects_world(4)
m, q0, dq0 = flatten_world(w)
bw = BatchedWorlds(m)
B = 512
q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1))
for skip in (False, True):
    for T in (40,):
        ts = []
        for _ in range(3):
            tq, tdq = bw.to_device(q, dq, torch.float32); cf = bw.new_cforce(B, torch.float32)
            torch.cuda.synchronize(); a = time.perf_counter()
            bw.step(tq, tdq, 5e-3, T, cforce=None if skip else cf, skip_constraints=skip); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
        print("skip_constraints", skip, "%.2f ms per step" % (min(ts) * 1e3 / T))
