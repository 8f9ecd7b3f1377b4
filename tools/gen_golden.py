#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference, imported through
tools/refload.py).  Only numbers are written: flattened models, inputs and the
reference's outputs.  The oracle (oracle/arb_oracle.py) and the HIP library are
both checked against these files; nothing here is imported by the product.

Usage:  python tools/gen_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402

arboris = refload.load()
from arboris.core import World, Body, simulate  # noqa: E402
from arboris.robots.human36 import add_human36  # noqa: E402
from arboris.robots.simpleshapes import add_groundplane  # noqa: E402
from arboris.robots.simplearm import add_simplearm  # noqa: E402
from arboris.robots.snake import add_snake  # noqa: E402
from arboris.controllers import WeightController, ProportionalDerivativeController  # noqa: E402
from arboris.constraints import (get_all_contacts, JointLimits,  # noqa: E402
                                 BallAndSocketConstraint, SoftFingerContact)
from arboris.joints import *  # noqa: E402,F401,F403
import arboris.joints as RJ  # noqa: E402
import arboris.homogeneousmatrix as Hg  # noqa: E402
import arboris.twistvector as TW  # noqa: E402
import arboris.collisions as COL  # noqa: E402

from arboris_python_amd.flatten import flatten_world, JT_FREE  # noqa: E402
from arboris_python_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FOUR = ('Right foot toe tip', 'Right foot heel', 'Left foot toe tip', 'Left foot heel')


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KiB" % (name, os.path.getsize(path) / 1024.))


def save_model(name, world):
    m, q, dq = flatten_world(world)
    d = m.to_npz_dict()
    d["q0"] = q
    d["dq0"] = dq
    save("model_%s.npz" % name, **d)
    return m


# ---- driving the reference at a given flat state ---------------------------
def set_state(world, m, q, dq):
    joints = list(world.iterjoints())
    for b, j in enumerate(joints):
        qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
        if m.jtype[b] == JT_FREE:
            j.gpos = q[qs].reshape(4, 4).copy()
        else:
            j.gpos[:] = q[qs]
    world._gvel[:] = dq


def get_state(world, m):
    q = np.concatenate([np.asarray(j.gpos, float).ravel() for j in world.iterjoints()])
    return q, world._gvel.copy()


def ref_step(world, dt):
    world.update_dynamic()
    world.update_controllers(dt)
    world.update_constraints(dt)
    world.integrate(dt)


def human36_ref(contacts=0):
    w = World()
    if contacts:
        add_groundplane(w)
    add_human36(w)
    w.register(WeightController())
    if contacts:
        for c in get_all_contacts(w, friction_coeff=.6):
            if contacts == 8 or c._shapes[1].name in FOUR:
                w.register(c)
    w.init()
    return w


# ---- G0: primitives --------------------------------------------------------
def gen_primitives():
    rng = np.random.default_rng(100)
    out = {}
    H = []
    for _ in range(8):
        H.append(Hg.transl(*rng.uniform(-2, 2, 3)) @ Hg.rotzyx(*rng.uniform(-3, 3, 3)))
    H = np.array(H)
    out["H"] = H
    out["H_inv"] = np.array([Hg.inv(h) for h in H])
    out["H_adjoint"] = np.array([Hg.adjoint(h) for h in H])
    out["H_iadjoint"] = np.array([Hg.iadjoint(h) for h in H])
    tw = rng.uniform(-3, 3, (8, 6))
    tw[6, 0:3] *= 1e-4          # below the 1e-3 series switch
    tw[7, 0:3] = 0.
    out["tw"] = tw
    out["tw_adjacency"] = np.array([TW.adjacency(t) for t in tw])
    out["tw_exp"] = np.array([TW.exp(t) for t in tw])
    vecs = [(1., 0., 0.), (0., 1., 0.), (0., 0., 1.), (0., -1., 0.)]
    for _ in range(6):
        v = rng.normal(size=3)
        vecs.append(tuple(v / np.linalg.norm(v)))
    out["zvec"] = np.array(vecs)
    out["zaligned"] = np.array([Hg.zaligned(np.array(v)) for v in vecs])
    # every joint type at random (q, dq)
    classes = ["FreeJoint", "RzRyRxJoint", "RzRyJoint", "RzRxJoint", "RyRxJoint",
               "RzJoint", "RyJoint", "RxJoint", "TxTyTzJoint"]
    for tid, cname in enumerate(classes):
        cls = getattr(RJ, cname)
        qs, dqs, poses, iposes, jacs, djacs, twists, idads = [], [], [], [], [], [], [], []
        for _ in range(4):
            if cname == "FreeJoint":
                q = Hg.transl(*rng.uniform(-1, 1, 3)) @ Hg.rotzyx(*rng.uniform(-3, 3, 3))
                dq = rng.uniform(-3, 3, 6)
                j = cls(gpos=q, gvel=dq)
            else:
                k = cls().ndof
                q = rng.uniform(-2, 2, k)
                dq = rng.uniform(-3, 3, k)
                j = cls(gpos=q, gvel=dq)
            qs.append(np.asarray(q).ravel()); dqs.append(dq)
            poses.append(j.pose); iposes.append(j.ipose); jacs.append(j.jacobian)
            djacs.append(j.djacobian); twists.append(j.twist); idads.append(j.idadjoint)
        out["joint%d_q" % tid] = np.array(qs)
        out["joint%d_dq" % tid] = np.array(dqs)
        out["joint%d_pose" % tid] = np.array(poses)
        out["joint%d_ipose" % tid] = np.array(iposes)
        out["joint%d_jac" % tid] = np.array(jacs)
        out["joint%d_djac" % tid] = np.array(djacs)
        out["joint%d_twist" % tid] = np.array(twists)
        out["joint%d_idadjoint" % tid] = np.array(idads)
    # plane/sphere collisions (collisions.py:161-205), incl. the doctest case
    pts = np.vstack(([2., 4., 3.], rng.uniform(-1, 1, (5, 3))))
    coeffs = np.array([0., 1., 0., -5.])
    res = [COL._plane_sphere_collision(np.eye(4), coeffs, p, 0.1) for p in pts]
    out["ps_points"] = pts
    out["ps_coeffs"] = coeffs
    out["ps_sdist"] = np.array([r[0] for r in res])
    out["ps_Hgc0"] = np.array([r[1] for r in res])
    out["ps_Hgc1"] = np.array([r[2] for r in res])
    save("g0_primitives.npz", **out)


# ---- G1: simplearm ---------------------------------------------------------
def read_h5_payload(fname):
    raw = open(os.path.join(refload.REFERENCE_ROOT, "tests", fname), "rb").read()
    return np.frombuffer(raw[:len(raw) // 8 * 8], "<f8")


def gen_simplearm():
    out = {}
    w = World()
    add_simplearm(w)
    m = save_model("simplearm", w)
    # tests/test_update_dynamic.py:10-19 state
    q = np.array([0.5, 1.0, 2.0 / 3.0])
    dq = np.array([2.5, -1.0, -0.5])
    set_state(w, m, q, dq)
    w.update_dynamic()
    bodies = list(w.ground.iter_descendant_bodies())
    out["ud_q"], out["ud_dq"] = q, dq
    out["ud_pose"] = np.array([b.pose for b in bodies])
    out["ud_jac"] = np.array([b.jacobian for b in bodies])
    out["ud_djac"] = np.array([b.djacobian for b in bodies])
    out["ud_twist"] = np.array([b.twist for b in bodies])
    out["ud_nle"] = np.array([b.nleffects for b in bodies])
    out["ud_M"], out["ud_B"], out["ud_N"] = w.mass.copy(), w.viscosity.copy(), w.nleffects.copy()
    # known answers printed in tests/test_update_dynamic.py:117-120, 195-198
    out["ud_M_known"] = np.array([[0.55132061, 0.1538999, 0.0080032],
                                  [0.1538999, 0.09002086, 0.00896043],
                                  [0.0080032, 0.00896043, 0.00267333]])
    out["ud_N_known"] = np.array([[0.11838112, -0.15894538, -0.01490104],
                                  [0.27979997, 0.00247348, -0.00494696],
                                  [0.03230564, 0.00742044, 0.]])
    # core.py:744-761 doctest: simplearm + PD(kp=2 on Elbow), dt=1e-3
    w = World()
    add_simplearm(w)
    joints = w.getjoints()
    w.register(ProportionalDerivativeController(joints[1:2], 2.))
    w.init()
    save_model("simplearm_pd", w)
    w.update_dynamic()
    w.update_controllers(0.001)
    out["pd_impedance"], out["pd_admittance"] = w._impedance.copy(), w._admittance.copy()
    out["pd_impedance_known"] = np.array([[686.98833333, 223.44666667, 20.67333333],
                                          [223.44666667, 93.44866667, 10.67333333],
                                          [20.67333333, 10.67333333, 2.67333333]])
    out["pd_admittance_known"] = np.array([[0.00732382, -0.0203006, 0.0244142],
                                           [-0.0203006, 0.07594182, -0.14621124],
                                           [0.0244142, -0.14621124, 0.76901683]])
    # config 1: gravity swing, timeline arange(0,1,.01) -> 99 steps
    # (tests/test_visu_collada.py:12-27 produced simplearm_*.h5 this way)
    w = World()
    w.register(WeightController())
    add_simplearm(w, with_shapes=True)
    w.getjoints()['Shoulder'].gpos[0] = 3.14 / 4
    m = save_model("simplearm_g", w)
    timeline = np.arange(0, 1, .01)
    w._current_time = timeline[0]
    w.init()
    bodies = list(w.ground.iter_descendant_bodies())
    names = [b.name for b in bodies]
    qs, dqs, poses, jposes = [], [], [], []
    for t_next in timeline[1:]:
        dt = t_next - w._current_time
        w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
        q_, dq_ = get_state(w, m)
        qs.append(q_); dqs.append(dq_)
        poses.append(np.array([b.pose for b in bodies]))
        jposes.append(np.array([b.parentjoint.pose for b in bodies]))
        w.integrate(dt)
    q_, dq_ = get_state(w, m)
    out["traj_timeline"] = timeline
    out["traj_q"], out["traj_dq"] = np.array(qs), np.array(dqs)
    out["traj_q_final"], out["traj_dq_final"] = q_, dq_
    out["traj_pose"] = np.array(poses)          # (99, 3, 4, 4) Arm, Forearm, Hand
    out["traj_jpose"] = np.array(jposes)
    out["traj_body_names"] = np.array(names)
    # raw float64 payload of the reference's golden HDF5 files:
    for flat in ("flat", "notflat"):
        a = read_h5_payload("simplearm_%s.h5" % flat)
        out["h5_%s_timeline" % flat] = a[406:505].copy()
        out["h5_%s_HandArmForearm" % flat] = a[505:505 + 3 * 99 * 16].reshape(3, 99, 4, 4).copy()
    save("g1_simplearm.npz", **out)


# ---- G2: human36 without contacts -------------------------------------------
def gen_human36():
    out = {}
    w = human36_ref(0)
    m = save_model("human36_g", w)
    # tests/test_human36.rst:93-113 known diagonal entries at q = 0
    w.update_dynamic()
    out["mass_diag_idx"] = np.array([5, 41, 40, 39, 16, 17, 10, 11])
    out["mass_diag_known"] = np.array([73.000000000000014, 0.10208399155688053,
                                       0.020356790291165189, 0.10430013572386694,
                                       0.0093741757009949949, 0.001397215796713388,
                                       0.0093741757009949949, 0.001397215796713388])
    out["mass_q0"] = w.mass.copy()
    B = 32
    q, dq = synth.random_states(m, B, seed=0)
    dts = np.where(np.arange(B) % 2 == 0, 5e-3, 1e-3)
    qn, dqn, Ms, Ns, Zs, gfs = [], [], [], [], [], []
    for i in range(B):
        set_state(w, m, q[i], dq[i])
        w.update_dynamic(); w.update_controllers(dts[i])
        if i < 4:
            Ms.append(w.mass.copy()); Ns.append(w.nleffects.copy())
            Zs.append(w._impedance.copy()); gfs.append(w._gforce.copy())
        w.update_constraints(dts[i]); w.integrate(dts[i])
        a, b = get_state(w, m)
        qn.append(a); dqn.append(b)
    out.update(q=q, dq=dq, dt=dts, q_next=np.array(qn), dq_next=np.array(dqn),
               M=np.array(Ms), N=np.array(Ns), Z=np.array(Zs), gforce=np.array(gfs))
    # 32-step rollouts of 4 states, dt = 5e-3 (milder velocities: with U(-3,3)
    # the free-flying humanoid diverges within 32 steps and parity is moot)
    q, dq = synth.random_states(m, 4, seed=1, vel=0.5)
    out["roll32_q0"], out["roll32_dq0"] = q, dq
    rq, rdq = [], []
    for i in range(4):
        set_state(w, m, q[i], dq[i])
        for _ in range(32):
            ref_step(w, 5e-3)
        a, b = get_state(w, m)
        rq.append(a); rdq.append(b)
    out["roll32_q"], out["roll32_dq"] = np.array(rq), np.array(rdq)
    save("g2_human36.npz", **out)


# ---- G3: human36 with floor contacts ------------------------------------------
def gen_contacts():
    out = {}
    captured = []
    orig_solve = SoftFingerContact.solve

    def spy(self, vel, admittance, dt):
        f0 = self._force.copy()
        df = orig_solve(self, vel, admittance, dt)
        captured.append((vel.copy(), admittance.copy(), f0, float(self._sdist), float(dt),
                         float(self._mu), np.array(df).copy(), self._force.copy()))
        return df
    SoftFingerContact.solve = spy
    for nc in (8, 4):
        w = human36_ref(nc)
        m = save_model("human36_c%d" % nc, w)
        # the reference's drop scenario (tests/test_human36_falling.py:10-45)
        root = w.ground.childrenjoints[0]
        root.gpos = np.dot(Hg.transl(0, 0.03, 0), root.gpos)
        dt = 5e-3
        qs, dqs, act, sd, frc = [], [], [], [], []
        cons = list(w._constraints)
        for step in range(39):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
            act.append([bool(c.is_active()) for c in cons])
            sd.append([c._sdist for c in cons])
            frc.append([c._force.copy() for c in cons])
            w.integrate(dt)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        out["drop%d_q" % nc] = np.array(qs)        # (40, nq): state before each step + final
        out["drop%d_dq" % nc] = np.array(dqs)
        out["drop%d_active" % nc] = np.array(act)
        out["drop%d_sdist" % nc] = np.array(sd)
        out["drop%d_force" % nc] = np.array(frc)
        out["drop%d_contact_height" % nc] = np.array(
            [(c._frames[1].pose[1, 3]) for c in cons])
        # random near-ground single steps: config-3 distribution with larger velocities
        B = 16
        q, dq = synth.standing_states(m, B, seed=3, drop=0.03, vel=0.5)
        # give half of them a tangential push so that sliding occurs
        dq[::2, 3] += 1.0
        q[:, 7] -= 0.035       # root y (4x4 entry [1,3]) so that feet are at/below the floor
        qn, dqn, fs = [], [], []
        for i in range(B):
            set_state(w, m, q[i], dq[i])
            ref_step(w, dt)
            a, b = get_state(w, m)
            qn.append(a); dqn.append(b)
            fs.append([c._force.copy() for c in cons])
        out["rand%d_q" % nc], out["rand%d_dq" % nc] = q, dq
        out["rand%d_q_next" % nc], out["rand%d_dq_next" % nc] = np.array(qn), np.array(dqn)
        out["rand%d_force" % nc] = np.array(fs)
    SoftFingerContact.solve = orig_solve
    # captured solve() tuples: keep up to 80 per branch
    rel, sta, sli = [], [], []
    for t in captured:
        (vel, adm, f0, sdist, dt, mu, df, f1) = t
        v0 = vel - adm @ f0
        if sdist + dt * v0[3] > 0:
            rel.append(t)
        elif np.allclose(df, -np.linalg.pinv(adm) @ np.hstack((vel[0:3], vel[3] + sdist / dt))):
            sta.append(t)
        else:
            sli.append(t)
    print("captured solves: release %d static %d sliding %d" % (len(rel), len(sta), len(sli)))
    rng = np.random.default_rng(5)
    for name, lst in (("release", rel), ("static", sta), ("sliding", sli)):
        if len(lst) > 80:
            idx = np.sort(rng.choice(len(lst), 80, replace=False))
            lst = [lst[i] for i in idx]
        out["solve_%s_vel" % name] = np.array([t[0] for t in lst])
        out["solve_%s_adm" % name] = np.array([t[1] for t in lst])
        out["solve_%s_force" % name] = np.array([t[2] for t in lst])
        out["solve_%s_sdist" % name] = np.array([t[3] for t in lst])
        out["solve_%s_dt" % name] = np.array([t[4] for t in lst])
        out["solve_%s_mu" % name] = np.array([t[5] for t in lst])
        out["solve_%s_dforce" % name] = np.array([t[6] for t in lst])
        out["solve_%s_newforce" % name] = np.array([t[7] for t in lst])
    save("g3_contacts.npz", **out)


# ---- G4: snake-64 -----------------------------------------------------------
def gen_snake():
    out = {}
    w = World()
    add_snake(w, 64)
    w.register(WeightController())
    w.init()
    m = save_model("snake64_g", w)
    B = 8
    q, dq = synth.random_states(m, B, seed=4, angle=0.5, vel=1.0)
    dt = 1e-3
    qn, dqn = [], []
    for i in range(B):
        set_state(w, m, q[i], dq[i])
        ref_step(w, dt)
        a, b = get_state(w, m)
        qn.append(a); dqn.append(b)
    out.update(q=q, dq=dq, dt=np.array(dt), q_next=np.array(qn), dq_next=np.array(dqn))
    set_state(w, m, q[0], dq[0])
    w.update_dynamic(); w.update_controllers(dt)
    out["Z0"] = w._impedance.copy()
    rq, rdq = [], []
    for i in range(2):
        set_state(w, m, q[i], dq[i])
        for _ in range(10):
            ref_step(w, dt)
        a, b = get_state(w, m)
        rq.append(a); rdq.append(b)
    out["roll10_q"], out["roll10_dq"] = np.array(rq), np.array(rdq)
    save("g4_snake64.npz", **out)


# ---- G5: energy drift (tests/test_energy_drift.py + energy_drift.h5) -----------
def gen_energy():
    out = {}
    a = read_h5_payload("energy_drift.h5")
    out["h5_kinetic_energy"] = a[-415:].copy()
    njoints = 9
    lengths = [1., .9, .8, .7, .6, .5, .4, .3, .2]
    masses = [1., .9, .8, .7, .6, .5, .4, .3, .2]
    gpos = [0., 3.14159 / 4., 0., 0., 0., 0., 0., 0., 0.]
    gvel = [2.] * njoints
    w = World()
    add_snake(w, njoints, lengths=lengths, masses=masses, gpos=gpos, gvel=gvel, is_fixed=False)
    w.register(WeightController())
    w.init()
    m = save_model("snake9_free_g", w)
    # emulate the integer-gpos truncation of 2010-era NumPy: the
    # joints built with an int gpos (all hinges but #1) never move.
    frozen = [b for b in range(m.nb) if m.jtype[b] != JT_FREE and b != 2]
    timeline = np.arange(0, 2.08, 0.005)
    ke = []
    joints = list(w.iterjoints())
    for t_next in timeline[1:]:
        dt = t_next - w._current_time
        w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
        ke.append(0.5 * w._gvel @ w.mass @ w._gvel)
        w.integrate(dt)
        for b in frozen:
            joints[b].gpos[:] = 0.
    out["ke_quirk"] = np.array(ke)
    out["frozen_bodies"] = np.array(frozen)
    out["timeline"] = timeline
    print("energy drift: max rel err vs h5 payload = %.3e"
          % np.max(np.abs(out["ke_quirk"] / out["h5_kinetic_energy"] - 1)))
    save("g5_energy.npz", **out)


# ---- G6: BallAndSocket / JointLimits (tests/test_constraints.py) ----------------
def gen_constraints():
    out = {}
    b0 = Body(mass=np.eye(6))
    w = World()
    w.add_link(w.ground, FreeJoint(), b0)
    w.init()
    w.register(WeightController())
    c0 = BallAndSocketConstraint(frames=(w.ground, b0))
    w.register(c0)
    w.init()
    m = save_model("ballsocket", w)
    dt = 0.001
    fr, qs, dqs = [], [], []
    for _ in range(5):
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
        fr.append(c0._force.copy())
        w.integrate(dt)
    a, b = get_state(w, m)
    qs.append(a); dqs.append(b)
    out["bs_force"] = np.array(fr)
    out["bs_q"], out["bs_dq"] = np.array(qs), np.array(dqs)
    out["bs_force_known"] = np.array([0., 9.81, 0.])          # tests/test_constraints.py:53
    # a ball-and-socket pendulum: arm hand pinned to the ground point where it starts
    for sign, tag in ((1., "max"), (-1., "min")):
        w = World()
        add_simplearm(w)
        w.register(WeightController())
        sh = w.getjoints()['Shoulder']
        w.register(JointLimits(sh, -3.14 / 2, 3.14 / 2))
        sh.gpos[0] = sign * (3.14 / 2 - 0.1)
        w.init()
        m = save_model("jointlimits_%s" % tag, w)
        qs, dqs = [], []
        for _ in range(99):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            ref_step(w, 1e-3)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        out["jl_%s_q" % tag], out["jl_%s_dq" % tag] = np.array(qs), np.array(dqs)
    save("g6_constraints.npz", **out)


# ---- G7: other narrow-phase pairs (collisions.py:113-159, 207-299) ----
from arboris_python_amd.scenes import shape_scenes   # scene definitions shared with the tests


class _RefNS(object):
    pass


def _ref_namespace():
    from arboris.core import SubFrame
    from arboris.shapes import Box, Sphere, Point
    from arboris.robots.simpleshapes import add_sphere
    import arboris.massmatrix as massmatrix
    W = _RefNS()
    W.World, W.Body, W.SubFrame, W.Hg = World, Body, SubFrame, Hg
    W.Box, W.Sphere, W.Point = Box, Sphere, Point
    W.add_sphere, W.add_groundplane = add_sphere, add_groundplane
    W.WeightController, W.get_all_contacts, W.FreeJoint = WeightController, get_all_contacts, FreeJoint
    W.massmatrix = massmatrix
    return W


def gen_shapes():
    W = _ref_namespace()
    out = {}
    dt = 5e-3
    for name, w in shape_scenes(W).items():
        m = save_model("shapes_" + name, w)
        cons = list(w._constraints)
        assert len(cons) >= 1, name
        qs, dqs, act, sd, frc = [], [], [], [], []
        for step in range(40):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
            act.append([bool(c.is_active()) for c in cons])
            sd.append([c._sdist for c in cons])
            frc.append([c._force.copy() for c in cons])
            w.integrate(dt)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        out[name + "_q"], out[name + "_dq"] = np.array(qs), np.array(dqs)
        out[name + "_active"], out[name + "_sdist"] = np.array(act), np.array(sd)
        out[name + "_force"] = np.array(frc)
        print("  %-12s contacts %d, active steps %d, max |f| %.2f" % (name, len(cons), int(np.array(act).any(1).sum()), np.abs(np.array(frc)).max()))
    save("g7_shapes.npz", **out)


# ---- G8: one ProportionalDerivativeController per world (controllers.py:63-158) ----
def gen_pd_per_world():
    """Each 'world' is a separate run of the reference with its own PD targets (and gains)."""
    from arboris.robots.simplearm import add_simplearm
    out = {}
    rng = np.random.RandomState(8)
    dt, nsteps = 5e-3, 30

    def arm(kp, kd, qdes, dqdes):
        w = World()
        add_simplearm(w)
        w.register(WeightController())
        js = [w.getjoints()[n] for n in ('Shoulder', 'Elbow', 'Wrist')]
        w.register(ProportionalDerivativeController(js, kp, kd, qdes, dqdes))
        w.init()
        js[0].gpos[0] = 0.4; js[1].gpos[0] = -0.3; js[2].gpos[0] = 0.2
        return w

    def run(w, m):
        qs, dqs = [], []
        for k in range(nsteps):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            ref_step(w, dt)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        return np.array(qs), np.array(dqs)

    # (a) shared full gain matrices, per-world targets
    A = rng.uniform(-1, 1, (3, 3)); KP = 30. * (A @ A.T + 3 * np.eye(3)) / 3.
    A = rng.uniform(-1, 1, (3, 3)); KD = 2. * (A @ A.T + 3 * np.eye(3)) / 3.
    W = 6
    qdes, dqdes = rng.uniform(-1, 1, (W, 3)), rng.uniform(-0.5, 0.5, (W, 3))
    Q, DQ = [], []
    for i in range(W):
        w = arm(KP, KD, qdes[i], dqdes[i])
        if i == 0:
            m = save_model("simplearm_pdw", w)
        a, b = run(w, m)
        Q.append(a); DQ.append(b)
    out["arm_t_q"], out["arm_t_dq"] = np.array(Q).transpose(1, 0, 2), np.array(DQ).transpose(1, 0, 2)   # (step, world, .)
    out["arm_t_qdes"], out["arm_t_dqdes"] = qdes, dqdes
    # (b) per-world diagonal gains and targets
    kp, kd = rng.uniform(5, 60, (W, 3)), rng.uniform(0.2, 4, (W, 3))
    qdes, dqdes = rng.uniform(-1, 1, (W, 3)), rng.uniform(-0.5, 0.5, (W, 3))
    Q, DQ = [], []
    for i in range(W):
        a, b = run(arm(np.diag(kp[i]), np.diag(kd[i]), qdes[i], dqdes[i]), m)
        Q.append(a); DQ.append(b)
    out["arm_g_q"], out["arm_g_dq"] = np.array(Q).transpose(1, 0, 2), np.array(DQ).transpose(1, 0, 2)
    out["arm_g_kp"], out["arm_g_kd"], out["arm_g_qdes"], out["arm_g_dqdes"] = kp, kd, qdes, dqdes
    # (c) human36 standing on 4 contacts, posture servo on every hinge, per-world posture targets
    W = 3
    n_h = None
    Q, DQ, QD, KPs, KDs = [], [], [], [], []
    for i in range(W):
        w = human36_ref(contacts=4)
        joints = [j for j in w.iterjoints() if not isinstance(j, FreeJoint)]
        dofs = np.concatenate([np.arange(j.dof.start, j.dof.stop) for j in joints])
        nd = len(dofs)
        tq = rng.uniform(-0.3, 0.3, nd)
        kpv, kdv = rng.uniform(100, 300, nd), rng.uniform(5, 20, nd)
        # the reference's PD.update stacks joint.gpos arrays (controllers.py:147-153), which needs
        # joints of equal ndof: one controller per joint size
        for size in sorted(set(j.ndof for j in joints)):
            grp = [j for j in joints if j.ndof == size]
            sel = np.concatenate([np.flatnonzero(dofs == d) for j in grp for d in range(j.dof.start, j.dof.stop)])
            w.register(ProportionalDerivativeController(grp, np.diag(kpv[sel]), np.diag(kdv[sel]), tq[sel],
                                                        np.zeros(len(sel))))
        w.init()
        if i == 0:
            mh = save_model("human36_c4_pdw", w)
        a, b = [], []
        for k in range(12):
            x, y = get_state(w, mh)
            a.append(x); b.append(y)
            ref_step(w, dt)
        x, y = get_state(w, mh)
        a.append(x); b.append(y)
        Q.append(np.array(a)); DQ.append(np.array(b))
        full = np.zeros(mh.ndof); fullp = np.zeros(mh.ndof); fulld = np.zeros(mh.ndof)
        full[dofs] = tq; fullp[dofs] = kpv; fulld[dofs] = kdv
        QD.append(full); KPs.append(fullp); KDs.append(fulld)
    out["h36_q"], out["h36_dq"] = np.array(Q).transpose(1, 0, 2), np.array(DQ).transpose(1, 0, 2)
    out["h36_qdes"], out["h36_kp"], out["h36_kd"] = np.array(QD), np.array(KPs), np.array(KDs)
    save("g8_pd_per_world.npz", **out)


# ---- G9: World.parse traversal order (core.py:562-606) ----
def gen_parse_order():
    from arboris_python_amd.exporters import ParseRecorder
    out = {}
    for name, w in (("human36_c8", human36_ref(contacts=8)), ("shapes_box_ball", shape_scenes(_ref_namespace())["box_ball"])):
        rec = ParseRecorder()
        w.parse(rec)
        out[name] = np.array(rec.calls)
        print("  %-16s %d hook calls" % (name, len(rec.calls)))
    save("g9_parse_order.npz", **out)


# ---- G10: body viscosity (core.py:729-731, 813) ----
def gen_viscosity():
    """human36 with a (non-symmetric) viscosity matrix on every body: world B matrix and 6 steps."""
    rng = np.random.RandomState(10)
    w = human36_ref(contacts=0)
    for b in w.iterbodies():
        if b is w.ground:
            continue
        A = rng.uniform(-1, 1, (6, 6))
        b.viscosity = 0.05 * (A @ A.T) + 0.02 * rng.uniform(-1, 1, (6, 6))
    w.init()
    m = save_model("human36_visc", w)
    q, dq = synth.random_states(m, 4, seed=21, vel=1.0)
    out = dict(q=q, dq=dq)
    Bs, qn, dqn = [], [], []
    for i in range(4):
        set_state(w, m, q[i], dq[i])
        w.update_dynamic()
        Bs.append(w.viscosity.copy())
        w.update_controllers(5e-3); w.update_constraints(5e-3); w.integrate(5e-3)
        a, b2 = get_state(w, m)
        qn.append(a); dqn.append(b2)
    out["B"], out["q_next"], out["dq_next"] = np.array(Bs), np.array(qn), np.array(dqn)
    save("g10_viscosity.npz", **out)


# ---- G11: TxTyTzJoint on the device path (joints.py:352-384) ----
def txtytz_world(W):
    """A gantry: ground -TxTyTz-> cart -RzRyRx-> arm -TxTyTz-> slider -Rx-> tip (+ a ball under the
    slider touching a ground plane), so that the prismatic joint occurs at the root and below rotating
    parents.  `W` is a namespace of classes (the reference's or this package's)."""
    Hg_ = W.Hg
    w = W.World()
    W.add_groundplane(w)
    mm = W.massmatrix
    cart = W.Body(name='cart', mass=mm.box((0.2, 0.1, 0.15), 2.0))
    arm = W.Body(name='arm', mass=mm.transport(mm.box((0.05, 0.3, 0.05), 1.2), Hg_.transl(0., 0.3, 0.)))
    slider = W.Body(name='slider', mass=mm.transport(mm.sphere(0.08, 0.7), Hg_.transl(0.02, -0.01, 0.03)))
    tip = W.Body(name='tip', mass=mm.transport(mm.cylinder(0.2, 0.03, 0.3), Hg_.transl(0., 0., 0.1)))
    f_cart = W.SubFrame(w.ground, Hg_.transl(0., 0.7, 0.), name='gantry base')
    w.add_link(f_cart, W.TxTyTzJoint(gpos=[0.1, 0.2, -0.1], gvel=[0.3, -0.2, 0.1], name='gantry'), cart)
    f_arm = W.SubFrame(cart, Hg_.transl(0.1, -0.1, 0.) @ Hg_.rotz(0.3), name='cart pivot')
    w.add_link(f_arm, W.RzRyRxJoint(gpos=[0.2, -0.4, 0.3], gvel=[0.5, 0.4, -0.6], name='pivot'), arm)
    f_sl = W.SubFrame(arm, Hg_.transl(0., -0.6, 0.) @ Hg_.rotx(0.2), name='arm end')
    f_sl1 = W.SubFrame(slider, Hg_.transl(0.01, 0.02, 0.) @ Hg_.roty(-0.1), name='slider mount')
    w.add_link(f_sl, W.TxTyTzJoint(gpos=[0.05, -0.02, 0.04], gvel=[-0.1, 0.2, 0.3], name='slide'), f_sl1)
    w.add_link(W.SubFrame(slider, Hg_.transl(0., -0.1, 0.)), W.RxJoint(gpos=[0.7], gvel=[-1.0], name='wrist'), tip)
    ball = W.SubFrame(slider, Hg_.transl(0., -0.12, 0.), name='slider ball frame')
    w.register(W.Sphere(ball, 0.05, name='slider ball'))
    w.register(W.WeightController())
    for c in W.get_all_contacts(w, friction_coeff=0.5):
        w.register(c)
    w.init()
    return w


def gen_txtytz():
    W = _ref_namespace()
    W.TxTyTzJoint, W.RzRyRxJoint, W.RxJoint = RJ.TxTyTzJoint, RJ.RzRyRxJoint, RJ.RxJoint
    w = txtytz_world(W)
    m = save_model("txtytz", w)
    assert 8 in list(m.jtype), m.jtype
    cons = list(w._constraints)
    out = {}
    dt = 5e-3
    # (a) rollout from the scene's own state: the ball reaches the floor, then contact
    qs, dqs, act, frc = [], [], [], []
    # (a sliding solve whose 6x6 matrix has complex eigenvalues makes the reference raise under NumPy 2 --
    # constraints.py:833 casts a complex-typed `s` with zero imaginary part into a float array, which
    # 2010-era NumPy allowed; the rollout stops there and such random states are skipped)
    for step in range(80):
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        w.update_dynamic(); w.update_controllers(dt)
        try:
            w.update_constraints(dt)
        except TypeError as e:
            print("  txtytz: rollout stops at step %d (%s)" % (step, type(e).__name__))
            qs.pop(); dqs.pop()
            break
        act.append([bool(c.is_active()) for c in cons]); frc.append([c._force.copy() for c in cons])
        w.integrate(dt)
    else:
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
    out["roll_q"], out["roll_dq"] = np.array(qs), np.array(dqs)
    out["roll_active"], out["roll_force"] = np.array(act), np.array(frc)
    print("  txtytz: contacts %d, active steps %d, max |f| %.2f" % (len(cons), int(np.array(act).any(1).sum()), np.abs(np.array(frc)).max()))
    # (b) random states, single steps + world matrices
    B = 24
    q, dq = synth.random_states(m, B, seed=11, angle=0.8, vel=2.0)
    qn, dqn, Ms, Ns = [], [], [], []
    nact = 0
    keep = np.ones(B, bool)
    for i in range(B):
        set_state(w, m, q[i], dq[i])
        w.update_dynamic()
        Ms.append(w.mass.copy()); Ns.append(w.nleffects.copy())
        try:
            w.update_controllers(dt); w.update_constraints(dt); w.integrate(dt)
        except TypeError:
            Ms.pop(); Ns.pop(); keep[i] = False
            continue
        nact += int(cons[0].is_active())
        a, b = get_state(w, m)
        qn.append(a); dqn.append(b)
    print("  txtytz: random states with an active contact: %d of %d" % (nact, B))
    q, dq = q[keep], dq[keep]
    out.update(q=q, dq=dq, q_next=np.array(qn), dq_next=np.array(dqn), M=np.array(Ms), N=np.array(Ns))
    save("g11_txtytz.npz", **out)


# ---- G12: rank-deficient constraint blocks -> numpy.linalg.pinv semantics (constraints.py:79, 83, 235, 795) ----
def gen_singular_blocks():
    """A planar 3R arm cannot move along z: the 3x3 admittance of a ball-and-socket on its hand has rank 2 and the
    4x4 admittance of a soft-finger contact rank <= 3.  The reference's pinv handles both; the goldens pin that."""
    from arboris.core import SubFrame
    from arboris.shapes import Plane, Point
    out = {}
    dt = 5e-3
    # (a) closed loop: the end effector pinned to the ground point where it starts (kinematic loop)
    w = World()
    add_simplearm(w)
    w.register(WeightController())
    js = w.getjoints()
    js['Shoulder'].gpos[0] = 0.9; js['Elbow'].gpos[0] = -1.1; js['Wrist'].gpos[0] = 0.6
    w.update_geometric()
    ee = w.getframes()['EndEffector']
    anchor = SubFrame(w.ground, Hg.transl(*ee.pose[0:3, 3]), name='anchor')
    c0 = BallAndSocketConstraint(frames=(anchor, ee))
    w.register(c0)
    w.init()
    m = save_model("loop_arm", w)
    qs, dqs, fr = [], [], []
    for _ in range(40):
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
        fr.append(c0._force.copy())
        w.integrate(dt)
    a, b = get_state(w, m)
    qs.append(a); dqs.append(b)
    out["loop_q"], out["loop_dq"], out["loop_force"] = np.array(qs), np.array(dqs), np.array(fr)
    w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
    Y = c0.jacobian @ w._admittance @ c0.jacobian.T
    sv = np.linalg.svd(Y, compute_uv=False)
    out["loop_block_singular_values"] = sv
    print("  loop_arm: singular values of the 3x3 block %s, max |f| %.2f" % (sv, np.abs(out["loop_force"]).max()))
    # (b) the arm comes down onto a plane: soft-finger contact at the end effector, once with a high friction
    #     coefficient (release and static branches) and once with a low one (sliding branch on a singular block)
    for tag, mu in (("contact_static", 3.0), ("contact_slide", 1.0)):
        w = World()
        w.register(Plane(w.ground, (0., 1., 0., -1.03), 'floor'))
        add_simplearm(w)
        w.register(WeightController())
        js = w.getjoints()
        js['Shoulder'].gpos[0] = 2.6; js['Elbow'].gpos[0] = 0.3; js['Wrist'].gpos[0] = 0.2
        ee = w.getframes()['EndEffector']
        w.register(Point(ee, 'tip'))
        cons = get_all_contacts(w, friction_coeff=mu)
        assert len(cons) == 1
        w.register(cons[0])
        w.init()
        m = save_model("planar_" + tag, w)
        qs, dqs, fr, act = [], [], [], []
        sv = None
        for step in range(120):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
            act.append(bool(cons[0].is_active())); fr.append(cons[0]._force.copy())
            if act[-1] and sv is None:
                Y = cons[0].jacobian @ w._admittance @ cons[0].jacobian.T
                sv = np.linalg.svd(Y, compute_uv=False)
            w.integrate(dt)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        out[tag + "_q"], out[tag + "_dq"] = np.array(qs), np.array(dqs)
        out[tag + "_force"], out[tag + "_active"] = np.array(fr), np.array(act)
        out[tag + "_block_singular_values"] = sv
        print("  planar %s: active steps %d, max |f| %.2f, singular values of the 4x4 block %s"
              % (tag, int(np.sum(act)), np.abs(out[tag + "_force"]).max(), sv))
    save("g12_singular.npz", **out)


# ---- G13: EnergyMonitor (observers.py:14-67; the module itself does not import under Python 3: TabError) ----
def gen_energy_monitor():
    """Kinetic and potential energy as EnergyMonitor.init/update compute them, statement by statement on the
    reference's objects with the reference's own `principalframe` (massmatrix.py:72-107), at random states."""
    from arboris.massmatrix import principalframe
    out = {}
    # (human36 is not in the list: `principalframe` asserts `ismassmatrix` and some of its bodies fail that test,
    # so the reference's EnergyMonitor cannot run on it)
    def arm():
        w = World(); add_simplearm(w); w.register(WeightController()); w.init()
        return w

    def snake():
        w = World(); add_snake(w, 9, is_fixed=False); w.register(WeightController()); w.init()
        return w
    for name, w in (("simplearm", arm()), ("snake9_free", snake())):
        m = save_model("energy_" + name, w)
        q, dq = synth.random_states(m, 8, seed=13, vel=1.5)
        bodies = list(w.ground.iter_descendant_bodies())
        com_pos = dict((b, principalframe(b.mass)[:, 3]) for b in bodies)          # observers.py:36-37
        ke, pe = [], []
        for i in range(len(q)):
            set_state(w, m, q[i], dq[i])
            w.update_dynamic()
            ke.append(np.dot(w.gvel, np.dot(w.mass, w.gvel)) / 2.)                  # observers.py:41-43
            Ep = 0.
            for b in bodies:                                                        # observers.py:45-49
                h = np.dot(np.dot(b.pose, com_pos[b])[0:3], w.up)
                Ep += b.mass[3, 3] * h
            pe.append(Ep * 9.81)
        out[name + "_q"], out[name + "_dq"] = q, dq
        out[name + "_ke"], out[name + "_pe"] = np.array(ke), np.array(pe)
    save("g13_energy_monitor.npz", **out)


def gen_user_controller():
    """g14 (round 6): a USER-DEFINED Controller with a dense impedance (tests/plugin_controllers.py, derived from the
    reference's own Controller ABC) registered on human36 with the four floor contacts; the reference's loop body run for
    12 steps from a state with the feet on the floor.  Recorded per step: the state, what the controller returned, the
    world's impedance and generalized force after update_controllers / update_constraints."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from plugin_controllers import make_spring_damper
    from arboris.core import Controller
    SpringDamper = make_spring_damper(Controller)
    w = World()
    add_groundplane(w)
    add_human36(w)
    w.register(WeightController())
    ctrl = SpringDamper()
    w.register(ctrl)
    for c in get_all_contacts(w, friction_coeff=.6):
        if c._shapes[1].name in FOUR:
            w.register(c)
    w.init()
    host = []
    m, _, _ = flatten_world(w, host_controllers=host)
    assert host == [ctrl]
    d = m.to_npz_dict()
    q, dq = synth.standing_states(m, 1, seed=21, drop=0.0, vel=0.1)
    q[:, 7] += 0.001                     # feet 1 mm above the floor (proximity 2 cm): the contacts are active from the first step
    rng = np.random.default_rng(22)
    for b in range(m.nb):                # bend the hinges a little: the spring has something to pull on
        if m.jtype[b] != JT_FREE:
            q[0, int(m.q_off[b]):int(m.q_off[b] + m.jnq[b])] = rng.uniform(-0.06, 0.06, int(m.jnq[b]))
    set_state(w, m, q[0], dq[0])
    dt, nsteps = 5e-3, 12
    qs, dqs, gfa, za, Z, gf0, gf1, frc = [], [], [], [], [], [], [], []
    orig = ctrl.update

    def spy(dt_):
        g, z = orig(dt_)
        gfa.append(np.array(g)); za.append(np.array(z))
        return g, z
    ctrl.update = spy
    cons = list(w._constraints)
    for k in range(nsteps):
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        w.update_dynamic()
        w.update_controllers(dt)
        Z.append(w._impedance.copy()); gf0.append(w._gforce.copy())
        w.update_constraints(dt)
        gf1.append(w._gforce.copy()); frc.append([c._force.copy() for c in cons])
        w.integrate(dt)
    a, b = get_state(w, m)
    qs.append(a); dqs.append(b)
    d.update(q=np.array(qs), dq=np.array(dqs), ctrl_gforce=np.array(gfa), ctrl_impedance=np.array(za),
             Z=np.array(Z), gforce0=np.array(gf0), gforce=np.array(gf1), cforce=np.array(frc), dt=dt)
    save("g14_user_controller.npz", **d)


def gen_wide():
    """g15 (round 6): worlds PAST 64 dofs run by the reference itself -- the pin of the wide kernels (csrc/arb_wide_kernel.h)
    and of the oracle at these sizes.  (a) add_snake(w, 100) under gravity: four random states, one step each, the world's
    impedance at the first, a 5-step rollout from two of them.  (b) human36 on the floor beside four free boxes carrying a
    ball each (66 dofs, 8 plane / sphere contacts: arboris_python_amd.scenes.human36_and_objects_world built from the
    reference's classes): 30 steps of the loop body from the scene's initial state with a small random velocity; per
    step the state, the active set, the constraint forces."""
    from arboris.shapes import Sphere
    from arboris.joints import FreeJoint
    import arboris.massmatrix as MM
    w = World()
    add_snake(w, 100)
    w.register(WeightController())
    w.init()
    m = save_model("snake100_g", w)
    out = {}
    B = 4
    q, dq = synth.random_states(m, B, seed=15, angle=0.5, vel=1.0)
    dt = 1e-3
    qn, dqn = [], []
    for i in range(B):
        set_state(w, m, q[i], dq[i])
        ref_step(w, dt)
        a, b = get_state(w, m)
        qn.append(a); dqn.append(b)
    out.update(snake_q=q, snake_dq=dq, snake_dt=np.array(dt), snake_q_next=np.array(qn), snake_dq_next=np.array(dqn))
    set_state(w, m, q[0], dq[0])
    w.update_dynamic(); w.update_controllers(dt)
    out["snake_Z0"] = w._impedance.copy()
    out["snake_gforce0"] = w._gforce.copy()
    rq, rdq = [], []
    for i in range(2):
        set_state(w, m, q[i], dq[i])
        for _ in range(5):
            ref_step(w, dt)
        a, b = get_state(w, m)
        rq.append(a); rdq.append(b)
    out["snake_roll5_q"], out["snake_roll5_dq"] = np.array(rq), np.array(rdq)

    for nobj, key, nsteps in ((4, "human", 30), (12, "human12", 16)):
        w = World()
        add_groundplane(w)
        add_human36(w)
        for k in range(nobj):
            he = (0.10 + 0.02 * k, 0.08, 0.12)
            body = Body(name="Box%d" % k, mass=MM.box(he, 2.0 + k))
            j = FreeJoint(name="BoxRoot%d" % k)
            j.gpos = Hg.transl(0.6 + 0.5 * k, 0.13 + 0.01 * k, 0.4 - 0.3 * k)
            w.add_link(w.ground, j, body)
            w.register(Sphere(body, 0.12, name="Box%d ball" % k))
        w.register(WeightController())
        for c in get_all_contacts(w, friction_coeff=.6):
            s0, s1 = c._shapes
            if type(s0).__name__ == "Plane" and (s1.name in FOUR or str(s1.name).endswith(" ball")):
                w.register(c)
        w.init()
        m = save_model("human36_obj%d" % nobj, w)
        assert m.ndof == 42 + 6 * nobj and m.nc == 4 + nobj, (m.ndof, m.nc)
        q0, dq0 = get_state(w, m)
        dq0 = dq0 + 0.05 * np.random.default_rng(151).standard_normal(m.ndof)
        set_state(w, m, q0, dq0)
        dt = 5e-3
        cons = list(w._constraints)
        qs, dqs, act, frc = [], [], [], []
        for k in range(nsteps):
            a, b = get_state(w, m)
            qs.append(a); dqs.append(b)
            w.update_dynamic(); w.update_controllers(dt); w.update_constraints(dt)
            act.append([bool(c.is_active()) for c in cons])
            frc.append([c._force.copy() for c in cons])
            w.integrate(dt)
        a, b = get_state(w, m)
        qs.append(a); dqs.append(b)
        out.update({key + "_q": np.array(qs), key + "_dq": np.array(dqs), key + "_active": np.array(act), key + "_force": np.array(frc), key + "_dt": np.array(dt)})
    save("g15_wide.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    table = dict(g0=gen_primitives, g1=gen_simplearm, g2=gen_human36, g3=gen_contacts,
                 g4=gen_snake, g5=gen_energy, g6=gen_constraints, g7=gen_shapes, g8=gen_pd_per_world, g9=gen_parse_order, g10=gen_viscosity, g11=gen_txtytz, g12=gen_singular_blocks, g13=gen_energy_monitor, g14=gen_user_controller, g15=gen_wide)
    for k in which:
        table[k]()
