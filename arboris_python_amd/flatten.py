"""Flatten an arboris ``World`` tree into the batch-shared model description.

The HIP step library (``include/arbstep.h``) and the CPU oracle (``oracle/``)
both consume the same flat, array-only description of a world: bodies in
depth-first order, one parent joint each, constant frames, 6x6 body matrices,
the contact table and the controller constants.  This is the counterpart of
the bookkeeping done by the reference in ``World.init`` (arboris/core.py:608-635:
DFS dof-slice assignment) and of everything its per-step code reads from the
object graph (``joint._frame0.bpose``, ``body.mass`` ...).

The walker is duck-typed on purpose: it only uses the public/protected
attributes that the reference classes and this package's classes share, and
dispatches on class *names*, so the golden-vector generator can flatten a world
built by the reference's own ``robots/*.py`` with the very same function.
"""
from __future__ import annotations

import numpy as np

# joint type ids shared with csrc/arb_kernels.hip (enum arb_joint_type)
JT_FREE, JT_RZRYRX, JT_RZRY, JT_RZRX, JT_RYRX, JT_RZ, JT_RY, JT_RX, JT_TXTYTZ = range(9)

JOINT_TYPES = {
    "FreeJoint": (JT_FREE, 6, 16),
    "RzRyRxJoint": (JT_RZRYRX, 3, 3),
    "RzRyJoint": (JT_RZRY, 2, 2),
    "RzRxJoint": (JT_RZRX, 2, 2),
    "RyRxJoint": (JT_RYRX, 2, 2),
    "RzJoint": (JT_RZ, 1, 1),
    "RyJoint": (JT_RY, 1, 1),
    "RxJoint": (JT_RX, 1, 1),
    "TxTyTzJoint": (JT_TXTYTZ, 3, 3),
}

# constraint type ids (enum arb_constraint_type)
CT_SOFTFINGER, CT_JOINTLIMITS, CT_BALLSOCKET = range(3)
CT_SOFTFINGER_PLANE = CT_SOFTFINGER      # historical name
# narrow-phase geometry of a SoftFingerContact (collisions.py): shape 0 / shape 1
CG_PLANE_SPHERE, CG_SPHERE_SPHERE, CG_BOX_SPHERE = range(3)


class UnsupportedModelError(NotImplementedError):
    """The world uses a plugin the batched device path cannot lower."""


def _class_names(obj):
    return [c.__name__ for c in type(obj).__mro__]


def _joint_type(joint):
    for name in _class_names(joint):
        if name in JOINT_TYPES:
            if name != type(joint).__name__:
                # a user subclass of a built-in joint may override pose/jacobian
                raise UnsupportedModelError(
                    "joint %r is a user-defined subclass (%s); only the built-in "
                    "joint types can be lowered to the device step"
                    % (getattr(joint, "name", None), type(joint).__name__))
            return JOINT_TYPES[name]
    raise UnsupportedModelError(
        "joint %r of type %s cannot be lowered to the device step"
        % (getattr(joint, "name", None), type(joint).__name__))


class FlatModel(object):
    """Array-only description of one world (shared by the whole batch)."""

    def __init__(self):
        self.nb = 0
        self.ndof = 0
        self.nq = 0
        self.body_names = []
        self.joint_names = []
        self.parent = np.zeros(0, np.int32)
        self.jtype = np.zeros(0, np.int32)
        self.dof_off = np.zeros(0, np.int32)
        self.jnd = np.zeros(0, np.int32)
        self.q_off = np.zeros(0, np.int32)
        self.jnq = np.zeros(0, np.int32)
        self.H_pr = np.zeros((0, 4, 4))
        self.H_cn = np.zeros((0, 4, 4))
        self.mass = np.zeros((0, 6, 6))
        self.visc = np.zeros((0, 6, 6))
        self.weighted = np.zeros(0, np.int32)   # bodies the WeightController acts on
        self.gravity = np.zeros(3)              # sum over WeightControllers of g*up
        self.up = np.array((0., 1., 0.))        # World.up
        # proportional-derivative controllers, merged, dof-indexed
        self.has_pd = False
        self.pd_kp = None
        self.pd_kd = None
        self.pd_tau0 = None                     # sum_a kp_a*qdes_a + kd_a*dqdes_a
        self.pd_mask = None                     # dofs some PD controller writes
        # constraints, registration order
        self.ctype = np.zeros(0, np.int32)
        self.c_names = []
        # SoftFingerContact: shape 0 (Plane | Sphere | Box) on body0 at bpose0, shape 1
        # (Sphere | Point) on body at c_local; bodies are -1 for the ground
        self.c_geom = np.zeros(0, np.int32)
        self.c_body = np.zeros(0, np.int32)
        self.c_local = np.zeros((0, 3))
        self.c_radius = np.zeros(0)
        self.c_radius0 = np.zeros(0)
        self.c_half = np.zeros((0, 3))
        self.c_plane = np.zeros((0, 4))
        self.c_mu = np.zeros(0)
        self.c_prox = np.zeros(0)
        self.c_eps = np.zeros((0, 3))
        self.c_enabled = np.zeros(0, np.int32)
        # JointLimits: dof index, min, max, proximity
        self.c_dof = np.zeros(0, np.int32)
        self.c_min = np.zeros(0)
        self.c_max = np.zeros(0)
        # BallAndSocket: (body0, bpose0, body1, bpose1), body -1 = ground
        self.c_body0 = np.zeros(0, np.int32)
        self.c_bpose0 = np.zeros((0, 4, 4))
        self.c_bpose1 = np.zeros((0, 4, 4))

    # -- convenience -------------------------------------------------------
    @property
    def nc(self):
        return len(self.ctype)

    def ancestors_dofs(self, b):
        """dof indices of the joints on the path ground -> body b (inclusive)."""
        dofs = []
        while b >= 0:
            dofs = list(range(self.dof_off[b], self.dof_off[b] + self.jnd[b])) + dofs
            b = int(self.parent[b])
        return dofs

    def to_npz_dict(self):
        d = {}
        for k, v in self.__dict__.items():
            if v is None:
                continue
            if isinstance(v, (list, tuple)):
                d[k] = np.array([("" if s is None else str(s)) for s in v], dtype="U96")
            else:
                d[k] = np.asarray(v)
        return d

    @classmethod
    def from_npz_dict(cls, d):
        m = cls()
        for k in d.keys():
            v = d[k]
            if k in ("body_names", "joint_names", "c_names"):
                setattr(m, k, [str(s) for s in v])
            elif v.shape == ():
                setattr(m, k, v.item())
            else:
                setattr(m, k, np.array(v))
        return m


def replicate_model(m, copies):
    """``copies`` independent instances of the flattened world ``m`` as ONE flattened world (a forest): copy k owns bodies
    ``k nb ..``, dofs ``k ndof ..``, position scalars ``k nq ..`` and constraints ``k nc ..``, so that a batch of states
    ``(B, nq)`` of ``m`` IS a batch ``(B / copies, copies nq)`` of the forest, without a copy.  The copies share ground,
    gravity and dt and nothing else: the impedance matrix is block diagonal, the Gauss-Seidel sweeps of one copy never
    see another's forces.  The host-side statement of what the LIBRARY does by itself for small models (``arb_model_create``
    builds the forest, ``BatchedWorlds.info["forest_copies"]`` says how many copies share a wavefront, ``step(..., one_world=True)``
    opts out): used by the tests and probes that build a forest by hand."""
    K = int(copies)
    if K < 1:
        raise ValueError("copies must be >= 1")
    f = FlatModel()
    f.nb, f.ndof, f.nq = K * m.nb, K * m.ndof, K * m.nq
    tag = lambda names: [("%s#%d" % (s, k) if K > 1 else s) for k in range(K) for s in names]
    f.body_names, f.joint_names, f.c_names = tag(m.body_names), tag(m.joint_names), tag(m.c_names)

    def tile(a, step=None):
        """concatenate K copies of ``a``; entries >= 0 of index arrays are shifted by k * step"""
        a = np.asarray(a)
        if step is None:
            return np.concatenate([a] * K, axis=0)
        return np.concatenate([np.where(a >= 0, a + k * step, a) for k in range(K)], axis=0).astype(a.dtype)

    f.parent = tile(m.parent, m.nb)
    f.jtype, f.jnd, f.jnq = tile(m.jtype), tile(m.jnd), tile(m.jnq)
    f.dof_off, f.q_off = tile(m.dof_off, m.ndof), tile(m.q_off, m.nq)
    f.dof2q = tile(m.dof2q, m.nq)
    f.H_pr, f.H_cn, f.mass, f.visc = tile(m.H_pr), tile(m.H_cn), tile(m.mass), tile(m.visc)
    f.weighted = tile(m.weighted)
    f.gravity, f.up = np.array(m.gravity, float), np.array(m.up, float)
    f.has_pd = bool(m.has_pd)
    if m.has_pd:
        eye = np.eye(K)
        f.pd_kp, f.pd_kd = np.kron(eye, m.pd_kp), np.kron(eye, m.pd_kd)
        f.pd_tau0, f.pd_mask = tile(m.pd_tau0), tile(m.pd_mask)
    f.ctype, f.c_geom, f.c_enabled = tile(m.ctype), tile(m.c_geom), tile(m.c_enabled)
    f.c_body, f.c_body0 = tile(m.c_body, m.nb), tile(m.c_body0, m.nb)
    f.c_dof = tile(m.c_dof, m.ndof)
    for name in ("c_local", "c_radius", "c_radius0", "c_half", "c_plane", "c_mu", "c_prox", "c_eps", "c_min", "c_max",
                 "c_bpose0", "c_bpose1"):
        setattr(f, name, tile(getattr(m, name)))
    return f


def _bpose(frame):
    return np.array(frame.bpose, dtype=np.float64).reshape(4, 4)


def flatten_world(world, positions=True, host_controllers=None):
    """Walk ``world`` depth-first and return ``(FlatModel, q0, dq0)``.

    ``host_controllers``: a list that receives the world's user-defined ``Controller`` objects (anything that is not one
    of the built-in classes, which are lowered to kernel constants).  The caller polls their ``update(dt)`` every step on
    the host and feeds the sums to the device as ``ext_gforce`` / ``ext_impedance`` (core.py:814-817; the object API does,
    ``_engine.py``).  Without the list such a controller raises ``UnsupportedModelError``.

    ``q0`` is the concatenation, in DFS joint order, of each joint's ``gpos``
    (ravelled; a FreeJoint contributes its 4x4 pose = 16 scalars, as stored by
    the reference in joints.py:30) and ``dq0`` of each joint's ``gvel``.
    DOF numbering follows arboris/core.py:611-615.
    """
    m = FlatModel()
    bodies = []          # DFS order, moving bodies only
    index_of = {}        # id(body) -> index
    parent, jtype, dof_off, jnd, q_off, jnq = [], [], [], [], [], []
    H_pr, H_cn, mass, visc = [], [], [], []
    q0, dq0 = [], []
    ndof = 0
    nq = 0

    # (depth-first without recursion: a chain of a thousand links is deeper than the interpreter's recursion limit)
    index_of[id(world.ground)] = -1
    stack = [(-1, iter(world.ground.childrenjoints))]
    while stack:
        body_index, joints = stack[-1]
        j = next(joints, None)
        if j is None:
            stack.pop()
            continue
        f0, f1 = j._frame0, j._frame1
        child = f1.body
        (tid, k, kq) = _joint_type(j)
        idx = len(bodies)
        bodies.append(child)
        index_of[id(child)] = idx
        parent.append(body_index)
        jtype.append(tid)
        dof_off.append(ndof)
        jnd.append(k)
        q_off.append(nq)
        jnq.append(kq)
        ndof += k
        nq += kq
        H_pr.append(_bpose(f0))
        H_cn.append(_bpose(f1))
        mass.append(np.array(child.mass, dtype=np.float64).reshape(6, 6))
        visc.append(np.array(child.viscosity, dtype=np.float64).reshape(6, 6))
        m.body_names.append(child.name)
        m.joint_names.append(j.name)
        q0.append(np.array(j.gpos, dtype=np.float64).ravel())
        dq0.append(np.array(j.gvel, dtype=np.float64).ravel())
        stack.append((idx, iter(child.childrenjoints)))

    m.nb, m.ndof, m.nq = len(bodies), ndof, nq
    m.up = np.array(world.up, dtype=float)
    m.parent = np.array(parent, np.int32)
    m.jtype = np.array(jtype, np.int32)
    m.dof_off = np.array(dof_off, np.int32)
    m.jnd = np.array(jnd, np.int32)
    m.q_off = np.array(q_off, np.int32)
    m.jnq = np.array(jnq, np.int32)
    m.H_pr = np.array(H_pr).reshape(-1, 4, 4)
    m.H_cn = np.array(H_cn).reshape(-1, 4, 4)
    m.mass = np.array(mass).reshape(-1, 6, 6)
    m.visc = np.array(visc).reshape(-1, 6, 6)

    # dof index -> position-scalar index for linear-configuration-space joints
    # (-1 for FreeJoint dofs, whose position is the 4x4 pose)
    dof2q = -np.ones(ndof, np.int32)
    for b in range(m.nb):
        if m.jtype[b] != JT_FREE:
            for i in range(m.jnd[b]):
                dof2q[m.dof_off[b] + i] = m.q_off[b] + i
    m.dof2q = dof2q

    # ---- controllers -----------------------------------------------------
    m.weighted = np.zeros(m.nb, np.int32)
    gravity = np.zeros(3)
    kp = np.zeros((ndof, ndof))
    kd = np.zeros((ndof, ndof))
    tau0 = np.zeros(ndof)
    pdmask = np.zeros(ndof, np.int32)
    n_weight = 0
    for a in getattr(world, "_controllers", []):
        names = _class_names(a)
        if "WeightController" in names and type(a).__name__ == "WeightController":
            # controllers.py:35-41: bodies with a non-zero mass matrix
            gravity = gravity + float(a.gravity) * np.asarray(world.up, float)
            n_weight += 1
            for b in range(m.nb):
                if np.linalg.norm(m.mass[b] > 0.):
                    m.weighted[b] = 1
        elif ("ProportionalDerivativeController" in names
              and type(a).__name__ == "ProportionalDerivativeController"):
            # controllers.py:141-158.  gforce[map] is *assigned* inside one
            # controller and summed across controllers; merging into full
            # dof-indexed matrices is exact as long as maps do not repeat a dof
            # inside one controller.
            dmap = []
            for j in a.joints:
                dmap.extend(range(j.dof.start, j.dof.stop))
            dmap = np.array(dmap, int)
            if len(set(dmap.tolist())) != len(dmap):
                raise UnsupportedModelError("PD controller lists a joint twice")
            m.has_pd = True
            kp_a = np.zeros((ndof, ndof))
            kd_a = np.zeros((ndof, ndof))
            kp_a[np.ix_(dmap, dmap)] = a.kp
            kd_a[np.ix_(dmap, dmap)] = a.kd
            # gforce_a = kp_a (qdes_a - q) + kd_a dqdes_a; summed over
            # controllers this is  tau0 - Kp q  with the constants below.
            kp += kp_a
            kd += kd_a
            qd = np.zeros(ndof)
            qd[dmap] = a.gpos_des
            dqd = np.zeros(ndof)
            dqd[dmap] = a.gvel_des
            tau0 += kp_a @ qd + kd_a @ dqd
            pdmask[dmap] = 1
        elif host_controllers is not None:
            host_controllers.append(a)
        else:
            raise UnsupportedModelError(
                "controller %r of type %s cannot be lowered to the device step: poll its update(dt) on the host and pass "
                "the (gforce, impedance) it returns as ext_gforce / ext_impedance (BatchedWorlds.step), or step the world "
                "through core.World / simulate, which does that" % (getattr(a, "name", None), type(a).__name__))
    if n_weight > 1:
        # each controller would add its own gravity; the sum is what we stored
        pass
    m.gravity = gravity
    if m.has_pd:
        m.pd_kp, m.pd_kd = kp, kd
        m.pd_tau0 = tau0
        m.pd_mask = pdmask

    # ---- constraints -----------------------------------------------------
    ctype, c_geom, c_body, c_local, c_radius, c_radius0, c_half, c_plane = [], [], [], [], [], [], [], []
    c_mu, c_prox, c_eps, c_enabled = [], [], [], []
    c_dof, c_min, c_max = [], [], []
    c_body0, c_bpose0, c_bpose1 = [], [], []
    I4 = np.eye(4)

    def _pad_common(c):
        m.c_names.append(getattr(c, "name", None))
        c_enabled.append(1 if c.is_enabled() else 0)

    def _defaults(skip=()):
        for lst, val in ((c_geom, 0), (c_body, -1), (c_local, np.zeros(3)), (c_radius, 0.), (c_radius0, 0.),
                         (c_half, np.zeros(3)), (c_plane, np.zeros(4)), (c_mu, 0.), (c_prox, 0.),
                         (c_eps, np.ones(3)), (c_dof, -1), (c_min, 0.), (c_max, 0.), (c_body0, -1),
                         (c_bpose0, I4), (c_bpose1, I4)):
            if not any(lst is x for x in skip):
                lst.append(val)

    for c in getattr(world, "_constraints", []):
        tname = type(c).__name__
        if tname == "SoftFingerContact":
            s0, s1 = c._shapes
            n0, n1 = type(s0).__name__, type(s1).__name__
            geom = {"Plane": CG_PLANE_SPHERE, "Sphere": CG_SPHERE_SPHERE, "Box": CG_BOX_SPHERE}.get(n0)
            if geom is None or n1 not in ("Point", "Sphere"):
                raise UnsupportedModelError(
                    "SoftFingerContact between %s and %s cannot be lowered (shape 0 must be a Plane, "
                    "Sphere or Box, shape 1 a Sphere or Point)" % (n0, n1))
            b0 = index_of.get(id(s0.frame.body), None)
            b1 = index_of.get(id(s1.frame.body), None)
            if b0 is None or b1 is None:
                raise UnsupportedModelError("contact shape on a body that is not in the world tree")
            ctype.append(CT_SOFTFINGER)
            c_geom.append(geom)
            c_body0.append(b0)
            c_bpose0.append(_bpose(s0.frame))
            c_body.append(b1)
            c_local.append(_bpose(s1.frame)[0:3, 3])
            c_radius.append(float(getattr(s1, "radius", 0.)))
            c_radius0.append(float(s0.radius) if geom == CG_SPHERE_SPHERE else 0.)
            c_half.append(np.array(s0.half_extents, float) if geom == CG_BOX_SPHERE else np.zeros(3))
            c_plane.append(np.array(s0.coeffs, float) if geom == CG_PLANE_SPHERE else np.zeros(4))
            c_mu.append(float(c._mu))
            c_prox.append(float(c._proximity))
            c_eps.append(np.array(c._eps, float))
            _defaults(skip=(c_geom, c_body0, c_bpose0, c_body, c_local, c_radius, c_radius0, c_half, c_plane,
                            c_mu, c_prox, c_eps))
            _pad_common(c)
        elif tname == "JointLimits":
            j = c._joint
            if j.ndof != 1:
                # constraints.py:67-71 compares 1-element arrays as booleans;
                # it only works for single-dof joints
                raise UnsupportedModelError("JointLimits on a multi-dof joint")
            ctype.append(CT_JOINTLIMITS)
            c_dof.append(int(j.dof.start))
            c_min.append(float(np.asarray(c._min).ravel()[0]))
            c_max.append(float(np.asarray(c._max).ravel()[0]))
            c_prox.append(float(np.asarray(c._proximity).ravel()[0]))
            _defaults(skip=(c_dof, c_min, c_max, c_prox))
            _pad_common(c)
        elif tname == "BallAndSocketConstraint":
            f0, f1 = c._frames
            ctype.append(CT_BALLSOCKET)
            c_body0.append(index_of[id(f0.body)])
            c_body.append(index_of[id(f1.body)])
            c_bpose0.append(_bpose(f0))
            c_bpose1.append(_bpose(f1))
            _defaults(skip=(c_body0, c_body, c_bpose0, c_bpose1))
            _pad_common(c)
        else:
            raise UnsupportedModelError(
                "constraint %r of type %s cannot be lowered to the device step"
                % (getattr(c, "name", None), tname))
    nc = len(ctype)
    m.ctype = np.array(ctype, np.int32)
    m.c_geom = np.array(c_geom, np.int32)
    m.c_body = np.array(c_body, np.int32)
    m.c_local = np.array(c_local, float).reshape(nc, 3)
    m.c_radius = np.array(c_radius, float)
    m.c_radius0 = np.array(c_radius0, float)
    m.c_half = np.array(c_half, float).reshape(nc, 3)
    m.c_plane = np.array(c_plane, float).reshape(nc, 4)
    m.c_mu = np.array(c_mu, float)
    m.c_prox = np.array(c_prox, float)
    m.c_eps = np.array(c_eps, float).reshape(nc, 3)
    m.c_enabled = np.array(c_enabled, np.int32)
    m.c_dof = np.array(c_dof, np.int32)
    m.c_min = np.array(c_min, float)
    m.c_max = np.array(c_max, float)
    m.c_body0 = np.array(c_body0, np.int32)
    m.c_bpose0 = np.array(c_bpose0, float).reshape(nc, 4, 4)
    m.c_bpose1 = np.array(c_bpose1, float).reshape(nc, 4, 4)

    q = np.concatenate(q0) if q0 else np.zeros(0)
    dq = np.concatenate(dq0) if dq0 else np.zeros(0)
    return m, q, dq
