"""Constraints solved by the world's Gauss-Seidel loop (host-side plugin API).

API mirror of arboris/constraints.py: JointLimits (:15-90),
BallAndSocketConstraint (:92-237), PointContact (:240-297), SoftFingerContact
(:300-836) and ``get_all_contacts`` (:839-878).

During stepping the built-in constraints are lowered by
``flatten.flatten_world`` to a constraint table and their ``update`` /
``jacobian`` / ``solve`` run inside the HIP kernel (``contact_update`` and
``gauss_seidel`` in csrc/arb_kernels.hip).  ``World.update_constraints`` writes
the device results back onto these objects (``_force``, ``_sdist``,
``_is_active``, contact frame poses).  ``solve`` -- the local solve of the
Gauss-Seidel sweeps, the hot part of the plugin API -- has ONE implementation in this
package, the device code of csrc/arb_math.h: called on its own (user code, API tests)
it runs that same code through the library's host build (``arb_host_softfinger_solve``,
``arb_host_block_pinv``), not a second NumPy restatement of the reference.
"""
import numpy as np
from numpy import array, zeros, dot

from . import homogeneousmatrix as Hg
from . import _capi
from .core import MovingSubFrame, Constraint, Shape, World


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _block_pinv(admittance):
    """pinv of a constraint's admittance block (numpy.linalg.pinv semantics) as the kernels compute it."""
    Y = _f64(np.atleast_2d(admittance))
    nd = Y.shape[0]
    P = np.zeros((nd, nd))
    if _capi.load().arb_host_block_pinv(_capi.ARB_F64, nd, _capi._dp(Y), _capi._dp(P)) < 0:
        raise ValueError("constraint blocks have 1 to 4 rows")
    return P

point_contact_proximity = 0.02
joint_limits_proximity = 0.01


def _rows(count):
    """`ndol` of a constraint class: a constant number of rows in the stacked constraint system."""
    return property(lambda self: count)


def _relative_jacobian(frames, rows):
    """Rows `rows` of the twist of frames[1] relative to frames[0], expressed in frames[0], as a function of gvel."""
    f0, f1 = frames
    carry = Hg.adjoint(dot(Hg.inv(f0.pose), f1.pose))
    return dot(carry[rows, :], f1.jacobian) - f0.jacobian[rows, :]


def _column(values, k):
    return array(values, dtype=float).reshape((k,))


class JointLimits(Constraint):
    """Keep ``min <= q <= max`` on a 1-dof joint with a unilateral force."""

    def __init__(self, joint, min, max, proximity=None, name=None):
        from .joints import LinearConfigurationSpaceJoint
        if not isinstance(joint, LinearConfigurationSpaceJoint):
            raise ValueError()
        Constraint.__init__(self, name)
        k = joint.ndof
        self._joint = joint
        self._min, self._max = _column(min, k), _column(max, k)
        self._proximity = (np.full(k, joint_limits_proximity) if proximity is None else _column(proximity, k))
        self._force = zeros(k)
        self._pos0 = self._jacobian = None

    ndol = _rows(1)
    jacobian = property(lambda self: self._jacobian)

    def init(self, world):
        selector = zeros((1, world.ndof))
        selector[0, self._joint.dof] = 1
        self._jacobian = selector

    def update(self, dt):
        self._force[:] = 0.
        self._pos0 = self._joint.gpos

    def is_active(self):
        return bool((self._pos0 - self._min < self._proximity)
                    or (self._max - self._pos0 < self._proximity))

    def solve(self, vel, admittance, dt):
        """Unilateral force that keeps the predicted position inside the limits (device rule, arb_kernels.hip
        gs_stage: with v0 the velocity without this force, push when v0 <= (min - q)/dt or (max - q)/dt <= v0)."""
        previous = self._force.copy()
        free_vel = np.asarray(vel, float) - dot(admittance, self._force)
        reach_min = (self._min - self._pos0) / dt
        reach_max = (self._max - self._pos0) / dt
        if free_vel <= reach_min:
            self._force = dot(_block_pinv(admittance), reach_min - free_vel)
        elif reach_max <= free_vel:
            self._force = dot(_block_pinv(admittance), reach_max - free_vel)
        else:
            self._force = zeros(previous.shape)
        return self._force - previous


class BallAndSocketConstraint(Constraint):
    """Keep the origins of two frames (on distinct bodies) coincident.

    Constraint velocity = linear velocity of frame 1 relative to frame 0 in
    frame 0; ``solve`` returns df = -Y^+ (v + p_01/dt).  The force is kept from
    one time step to the next (warm start).
    """

    def __init__(self, frames, name=None):
        Constraint.__init__(self, name)
        self._frames = frames
        self._force = zeros(3)
        self._pos0 = None

    ndol = _rows(3)
    jacobian = property(lambda self: _relative_jacobian(self._frames, slice(3, 6)))

    def init(self, world):
        pass

    def is_active(self):
        return True

    def update(self, dt):
        f0, f1 = self._frames
        self._pos0 = dot(Hg.inv(f0.pose), f1.pose)[0:3, 3]

    def solve(self, vel, admittance, dt):
        dforce = -dot(_block_pinv(admittance), np.asarray(vel, float) + self._pos0 / dt)
        self._force = self._force + dforce
        return dforce


class PointContact(Constraint):
    """Base of point contacts: collision detection + contact frame placement.

    ``update`` calls the pair's collision solver, moves the two contact frames
    (same orientation, z = normal) onto the bodies and activates the contact
    when the predicted gap ``sdist + dsdist*dt`` is below ``proximity``.
    """

    def __init__(self, shapes, collision_solver, proximity, name):
        assert isinstance(shapes[0], Shape)
        assert isinstance(shapes[1], Shape)
        Constraint.__init__(self, name)
        if collision_solver is None:
            from .collisions import choose_solver
            shapes, collision_solver = choose_solver(*shapes[:2])
        self._shapes, self._collision_solver, self._proximity = shapes, collision_solver, proximity
        # one moving frame per body, placed on the contact point by update()
        self._frames = tuple(MovingSubFrame(shape.frame.body) for shape in shapes[:2])
        self._is_active = self._sdist = None

    def init(self, world):
        pass

    def update(self, dt):
        sdist, *contact_poses = self._collision_solver(self._shapes)
        for frame, shape, H_gc in zip(self._frames, self._shapes, contact_poses):
            frame.bpose = dot(Hg.inv(shape.frame.body.pose), H_gc)
        # rate of change of the gap: z velocity of contact frame 1 relative to contact frame 0
        carry = Hg.adjoint(dot(Hg.inv(contact_poses[0]), contact_poses[1]))
        gap_rate = dot(carry[5, :], self._frames[1].twist) - self._frames[0].twist[5]
        self._sdist = sdist
        self._is_active = bool(sdist + dt * gap_rate < self._proximity)
        self._force[:] = 0.

    def is_active(self):
        return self._is_active


class SoftFingerContact(PointContact):
    """Point contact with elliptic Coulomb friction including a torsional term.

    Constraint space (4 rows): (w_z, v_x, v_y, v_z) of frame 1 relative to
    frame 0; force (m_z, f_x, f_y, f_z).  ``solve`` distinguishes release,
    static friction and sliding friction; the sliding branch reproduces the
    reference's arithmetic (its 1-D ``dot`` products are scalars).
    """

    def __init__(self, shapes, friction_coeff, collision_solver=None,
                 proximity=point_contact_proximity, name=None):
        self._mu = friction_coeff
        PointContact.__init__(self, shapes, collision_solver, proximity, name)
        self._force = zeros(4)
        self._eps = array((1., 1., 1.))

    ndol = _rows(4)
    jacobian = property(lambda self: _relative_jacobian(self._frames, slice(2, 6)))

    def solve(self, vel, admittance, dt):
        """Release / static friction / sliding friction (arb_math.h::softfinger_solve, the code the Gauss-Seidel
        kernels run, in float64); updates ``_force`` and returns the force increment."""
        vel, adm, force, eps = _f64(vel), _f64(admittance), _f64(self._force).copy(), _f64(self._eps)
        dforce = np.zeros(4)
        branch = _capi.load().arb_host_softfinger_solve(
            _capi.ARB_F64, _capi._dp(vel), _capi._dp(adm), _capi._dp(force), float(self._sdist), float(dt),
            float(self._mu), _capi._dp(eps), _capi._dp(dforce))
        if branch < 0:
            raise ValueError("SoftFingerContact.solve: bad arguments")
        self._force = force
        return dforce


def get_all_contacts(world, contact_class=None, **args):
    """One contact per pair of shapes on distinct bodies for which a collision
    solver exists; extra keyword arguments go to the contact constructor.
    The caller registers the returned contacts."""
    assert isinstance(world, World)
    if contact_class is None:
        contact_class = SoftFingerContact
    else:
        assert issubclass(contact_class, PointContact)
    shapes = tuple(world.itershapes())
    contacts = []
    for i, s0 in enumerate(shapes):
        for s1 in shapes[i + 1:]:
            if s0.frame.body is s1.frame.body:
                continue
            try:
                contacts.append(contact_class((s0, s1), **args))
            except NotImplementedError:
                pass
    return contacts
