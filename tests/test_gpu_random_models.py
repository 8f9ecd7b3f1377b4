"""Randomised models: trees the shipped robots never build -- every joint type anywhere in the tree, joints mounted
through rotated SubFrames on BOTH sides (H_pr and H_cn general, core.py:1295-1298), several roots, off-centre
inertias, viscosity, spheres touching a ground plane and one another, a ball-and-socket loop closure and joint
limits -- stepped on the device against the oracle (which restates core.py:1356-1363 for any flattened model and is
pinned to the reference on the golden scenes, among them g11's gantry with a general H_cn).
float64: 1e-8 per world -- the functional check.  float32 is a sanity check here, not the 1e-5 contract of the
BASELINE models: these trees mix 3 cm / 0.2 kg boxes with 20 cm / 3 kg ones, loop closures start violated by tens of
centimetres (|dq+| of 1e2..1e3 rad/s after one step), and the float32 solve of such graded, violently corrected
systems lands at 1e-5..2e-4 there: every world finite and below 1e-3; worlds with |dq+| < 30 rad/s below 1e-4 and 70 % of
them below 1e-5."""
import numpy as np
import pytest

import arb_oracle as O
from parity_tools import explain_outlier, ill_conditioned

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def random_world(seed, nbody_range=(3, 11), max_dof=40, max_contacts=6, max_spheres=4):
    from arboris_python_amd.core import World, Body, SubFrame
    from arboris_python_amd import joints as J, massmatrix as mm, homogeneousmatrix as Hg
    from arboris_python_amd.shapes import Sphere, Plane, Point
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.constraints import get_all_contacts, BallAndSocketConstraint, JointLimits
    rng = np.random.default_rng(seed)
    w = World()
    w.register(Plane(w.ground, (0., 1., 0., -0.4), 'floor'))
    kinds = [J.FreeJoint, J.RzRyRxJoint, J.RzRyJoint, J.RzRxJoint, J.RyRxJoint, J.RzJoint, J.RyJoint, J.RxJoint, J.TxTyTzJoint]
    nbody = int(rng.integers(*nbody_range))
    bodies = []

    def rand_frame(scale=0.3):
        return Hg.transl(*rng.uniform(-scale, scale, 3)) @ Hg.rotzyx(*rng.uniform(-1.2, 1.2, 3))
    hinges = []
    ndof = 0
    for k in range(nbody):
        he = rng.uniform(0.03, 0.2, 3)
        M = mm.transport(mm.box(he, float(rng.uniform(0.2, 3.))), Hg.transl(*rng.uniform(-0.1, 0.1, 3)))
        b = Body(name='b%d' % k, mass=M)
        if rng.uniform() < 0.4:
            A = rng.uniform(-1, 1, (6, 6))
            b.viscosity = 0.02 * (A @ A.T)
        parent = w.ground if (k == 0 or rng.uniform() < 0.15) else bodies[int(rng.integers(0, len(bodies)))]
        kind = kinds[int(rng.integers(0, len(kinds)))] if k else kinds[int(rng.choice([0, 1, 8, 5]))]
        if ndof + kind().ndof > max_dof:
            break
        j = kind(name='j%d' % k)
        if isinstance(j, J.FreeJoint):
            j.gpos = Hg.transl(*rng.uniform(-0.3, 0.3, 3)) @ Hg.rotzyx(*rng.uniform(-1, 1, 3))
        else:
            j.gpos[:] = rng.uniform(-0.8, 0.8, j.ndof)
        j.gvel[:] = rng.uniform(-1.5, 1.5, j.ndof)
        f0 = SubFrame(parent, rand_frame(), name='p%d' % k)
        f1 = SubFrame(b, rand_frame(0.1), name='c%d' % k) if rng.uniform() < 0.6 else b
        w.add_link(f0, j, f1)
        ndof += j.ndof
        bodies.append(b)
        if j.ndof == 1:
            hinges.append(j)
    nsph = 0
    for b in bodies:
        if rng.uniform() < 0.45 and nsph < max_spheres:
            fr = SubFrame(b, Hg.transl(*rng.uniform(-0.15, 0.15, 3)), name='s%d' % nsph)
            w.register(Sphere(fr, float(rng.uniform(0.03, 0.12)), name='ball%d' % nsph) if rng.uniform() < 0.7
                       else Point(fr, name='pt%d' % nsph))
            nsph += 1
    w.register(WeightController())
    ncon = 0
    for c in get_all_contacts(w, friction_coeff=float(rng.uniform(0.3, 1.2))):
        if ncon < max_contacts:
            w.register(c)
            ncon += 1
    if len(bodies) >= 3 and rng.uniform() < 0.5:
        w.register(BallAndSocketConstraint(frames=(SubFrame(bodies[0], rand_frame(0.1)), SubFrame(bodies[-1], rand_frame(0.1)))))
    if hinges and rng.uniform() < 0.6:
        jl = hinges[0]
        w.register(JointLimits(jl, float(jl.gpos[0]) - 0.005, float(jl.gpos[0]) + 0.5))
    w.init()
    return w


@pytest.mark.parametrize("seed", range(16))
def test_random_model_single_steps(seed):
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd import synth
    w = random_world(seed)
    m, q0, dq0 = flatten_world(w)
    assert m.ndof <= 64 and m.nc <= 16
    bw = BatchedWorlds(m)
    B = 12
    rng = np.random.default_rng(100 + seed)
    q = np.tile(q0, (B, 1)); dq = np.tile(dq0, (B, 1))
    qr, dqr = synth.random_states(m, B, seed=seed, angle=0.8, vel=1.5, root_box=((-.3, .3), (-.2, .5), (-.3, .3)))
    q[1:], dq[1:] = qr[1:], dqr[1:]                     # world 0 = the scene as built, the others random
    dt = float(rng.choice([2e-3, 5e-3]))
    cf0 = np.zeros((B, m.nc, 4))
    oq, odq, ocf = O.step(m, q, dq, dt, cforce=cf0)
    ok = np.isfinite(oq).all(axis=1) & np.isfinite(odq).all(axis=1) & (np.abs(odq).max(axis=1) < 1e3)
    assert ok.sum() >= 2                                 # (random states may start deep inside the floor or far from a loop closure)
    tq, tdq = bw.to_device(q, dq, torch.float64)
    tcf = bw.new_cforce(B, torch.float64)
    bw.step(tq, tdq, dt, 1, cforce=tcf)
    torch.cuda.synchronize()
    eq = np.abs(tq.cpu().numpy() - oq).max(axis=1) / np.maximum(1., np.abs(oq).max(axis=1))
    edq = np.abs(tdq.cpu().numpy() - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
    assert eq[ok].max() < 1e-8 and edq[ok].max() < 1e-7, (seed, eq[ok].max(), edq[ok].max(), list(m.jtype), list(m.ctype))
    # float32 on the float32-rounded inputs
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    oq32, odq32, _ = O.step(m, f(q), f(dq), dt, cforce=cf0)
    ok32 = ok & np.isfinite(oq32).all(axis=1) & (np.abs(odq32).max(axis=1) < 1e3)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(B, torch.float32)
    bw.step(sq, sdq, dt, 1, cforce=scf)
    torch.cuda.synchronize()
    e32 = np.maximum(np.abs(sq.cpu().numpy() - oq32).max(axis=1) / np.maximum(1., np.abs(oq32).max(axis=1)),
                     np.abs(sdq.cpu().numpy() - odq32).max(axis=1) / np.maximum(1., np.abs(odq32).max(axis=1)))
    assert np.isfinite(e32[ok32]).all() and e32[ok32].max() < 1e-3, (seed, e32[ok32])
    # Every world over the 1e-5 gate is adjudicated (round 4; until then: "70 % of the calm worlds below 1e-5"): a decision
    # of a SoftFingerContact solve that differs from the oracle's and is marginal for the oracle (parity_tools criteria
    # a-e), or identical decisions in a step so ill-conditioned that the float64 oracle itself moves by at least the
    # error under a one-ulp (float32) input change.  (BallAndSocket rows have no decision; the clamp decision of a
    # JointLimits solve is not in the device's trace -- code 4 -- nor in the oracle's, so a world whose only difference
    # is that clamp has to pass the conditioning test.)  Worlds torn towards a violated loop closure (|dq+| of 1e2 ... 1e3
    # rad/s) are held to 1e-3 as before, the calm ones to 1e-4.
    calm = ok32 & (np.abs(odq32).max(axis=1) < 30.)       # worlds that are not being torn towards a violated loop closure
    if calm.any():
        assert e32[calm].max() < 1e-4, (seed, e32[calm])
    eq32 = np.abs(sq.cpu().numpy() - oq32).max(axis=1) / np.maximum(1., np.abs(oq32).max(axis=1))
    edq32 = np.abs(sdq.cpu().numpy() - odq32).max(axis=1) / np.maximum(1., np.abs(odq32).max(axis=1))
    for wi in np.flatnonzero(calm & (e32 >= 1e-5)):
        qw, dqw = np.asarray(q[wi], np.float32), np.asarray(dq[wi], np.float32)
        why = explain_outlier(bw, m, qw, dqw, dt) if m.nc else None
        if why is None:
            why = ill_conditioned(m, qw, dqw, dt, eq32[wi], edq32[wi], cap=1e-4)
        assert why is not None, "seed %d world %d: float32 error q %.2e dq %.2e unexplained" % (seed, wi, eq32[wi], edq32[wi])
        print("seed %d world %d over the gate (q %.2e dq %.2e) [%s]: %s" % (seed, wi, eq32[wi], edq32[wi], why.criterion, why))
    # a multi-step launch of more worlds than wave slots goes through the work queue (whatever register tile the
    # model selects): bit-identical to one workgroup per world
    reps = -(-5000 // B)
    qb, dqb = np.tile(q, (reps, 1)), np.tile(dq, (reps, 1))
    for dtype in (torch.float32, torch.float64):
        res = []
        for static in (True, False):
            aq, adq = bw.to_device(qb, dqb, dtype)
            acf = bw.new_cforce(len(qb), dtype)
            bw.step(aq, adq, dt, 3, cforce=acf, static_worlds=static)
            res.append((aq, adq, acf))
        torch.cuda.synchronize()
        # (NaN-safe: compare bit patterns)
        bits = lambda t: t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int64)
        assert all(torch.equal(bits(a), bits(b)) for a, b in zip(*res)), (seed, dtype)
    # small models: the worlds of a large batch share wavefronts (the library's forest of k copies) -- bit for bit what
    # one world per wavefront computes, for every world that stays finite there; worlds that do not are retired (NaN)
    if bw.info["forest_copies"] > 1:
        for dtype in (torch.float32, torch.float64):
            res = []
            for one in (True, False):
                aq, adq = bw.to_device(qb, dqb, dtype)
                acf = bw.new_cforce(len(qb), dtype)
                assert bw.plan(len(qb), 3, dtype=dtype, one_world=one)["worlds_per_wavefront"] == (1 if one else bw.info["forest_copies"])
                bw.step(aq, adq, dt, 3, cforce=acf, one_world=one)
                res.append((aq.cpu().numpy(), adq.cpu().numpy(), acf.cpu().numpy().reshape(len(qb), -1)))
            torch.cuda.synchronize()
            lim = 1e8 if dtype == torch.float32 else 1e100
            fin = np.all(np.abs(res[0][0]) < lim, axis=1) & np.all(np.abs(res[0][1]) < lim, axis=1) & np.all(np.abs(res[0][2]) < lim, axis=1)
            assert fin.sum() >= len(qb) // 6
            # (bit for bit, with or without constraints: since round 4 the constraint-space products of phase D add a copy's
            # dofs in the groups of four they form in the copy alone -- until then models with constraints and ndof % 4 != 0
            # agreed to a few ulps only)
            for k, (a1, af) in enumerate(zip(res[0], res[1])):
                assert np.array_equal(a1[fin], af[fin]), (seed, dtype, bw.info["forest_copies"], k)
    bw.close()
