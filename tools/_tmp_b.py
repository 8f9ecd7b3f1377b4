import sys, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from arboris_python_amd.core import World, Body
from arboris_python_amd.joints import FreeJoint
from arboris_python_amd.shapes import Sphere
from arboris_python_amd import massmatrix, homogeneousmatrix as Hg
from arboris_python_amd.robots.human36 import add_human36
from arboris_python_amd.robots.simpleshapes import add_groundplane
from arboris_python_amd.controllers import WeightController
from arboris_python_amd.constraints import get_all_contacts
from arboris_python_amd.flatten import flatten_world
from arboris_python_amd.batch import BatchedWorlds
names = ["A", "A'", "B", "Z", "C", "D", "GS"]
for nballs in (3, 8):
    w = World(); add_groundplane(w); add_human36(w)
    for k in range(nballs):
        body = Body(name="Ball%d" % k, mass=massmatrix.sphere(0.1, 1.0 + k))
        j = FreeJoint(name="BallRoot%d" % k); j.gpos = Hg.transl(0.12 + 0.19 * (k % 4), 0.105, 0.05 * (k % 4) + 0.21 * (k // 4))
        w.add_link(w.ground, j, body); w.register(Sphere(body, 0.1, name="Ball%d" % k))
    w.register(WeightController())
    for c in get_all_contacts(w, friction_coeff=0.6): w.register(c)
    w.init()
    m, q0, dq0 = flatten_world(w)
    B = 256
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1)) + 0.05 * np.random.RandomState(1).standard_normal((B, m.ndof))
    bw = BatchedWorlds(m)
    tq, tdq = bw.to_device(q, dq, torch.float64); cf = bw.new_cforce(B, torch.float64)
    for k in range(20):
        if k in (0, 19):
            r = bw.inspect(tq, tdq, 5e-3, ["stamps", "c_active"], cforce=cf)
            st = r["stamps"].double(); d = (st[:, 1:] - st[:, :-1]).mean(0).tolist()
            print("human36 + %d balls (%d contacts, %.1f active) step %2d: " % (nballs, m.nc, float(r["c_active"].float().sum(1).mean()), k) + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, d)) + "   sum %.0f" % sum(d))
        bw.step(tq, tdq, 5e-3, 1, cforce=cf)
