"""Constraints solved by the world's Gauss-Seidel loop (host-side plugin API).

API mirror of arboris/constraints.py: JointLimits (:15-90),
BallAndSocketConstraint (:92-237), PointContact (:240-297), SoftFingerContact
(:300-836) and ``get_all_contacts`` (:839-878).

During stepping the built-in constraints are lowered by
``flatten.flatten_world`` to a constraint table and their ``update`` /
``jacobian`` / ``solve`` run inside the HIP kernel (``contact_update`` and
``gauss_seidel`` in csrc/arb_kernels.hip).  ``World.update_constraints`` writes
the device results back onto these objects (``_force``, ``_sdist``,
``_is_active``, contact frame poses).  The NumPy methods below implement the
same per-constraint algebra for user code and for tests of the API itself.
"""
import numpy as np
from numpy import array, zeros, eye, dot, hstack, diag
from numpy.linalg import solve, eigvals, pinv

from . import homogeneousmatrix as Hg
from .core import MovingSubFrame, Constraint, Shape, World

point_contact_proximity = 0.02
joint_limits_proximity = 0.01


class JointLimits(Constraint):
    """Keep ``min <= q <= max`` on a 1-dof joint with a unilateral force."""

    def __init__(self, joint, min, max, proximity=None, name=None):
        from .joints import LinearConfigurationSpaceJoint
        if not isinstance(joint, LinearConfigurationSpaceJoint):
            raise ValueError()
        Constraint.__init__(self, name)
        k = joint.ndof
        self._joint = joint
        self._min = array(min, dtype=float).reshape((k,))
        self._max = array(max, dtype=float).reshape((k,))
        if proximity is None:
            self._proximity = joint_limits_proximity * np.ones((k,))
        else:
            self._proximity = array(proximity, dtype=float).reshape((k,))
        self._pos0 = None
        self._jacobian = None
        self._force = zeros((k,))

    def init(self, world):
        self._jacobian = zeros((1, world.ndof))
        self._jacobian[0, self._joint.dof] = 1

    @property
    def jacobian(self):
        return self._jacobian

    @property
    def ndol(self):
        return 1

    def update(self, dt):
        self._pos0 = self._joint.gpos
        self._force[:] = 0.

    def is_active(self):
        return bool((self._pos0 - self._min < self._proximity)
                    or (self._max - self._pos0 < self._proximity))

    def solve(self, vel, admittance, dt):
        previous = self._force.copy()
        predicted = self._pos0 + dt * (vel - dot(admittance, self._force))
        if predicted <= self._min:
            self._force = dot(pinv(admittance), (self._min - predicted) / dt)
        elif self._max <= predicted:
            self._force = dot(pinv(admittance), (self._max - predicted) / dt)
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
        self._force = zeros(3)
        self._pos0 = None
        Constraint.__init__(self, name)
        self._frames = frames

    def init(self, world):
        pass

    @property
    def ndol(self):
        return 3

    def _h01(self):
        return dot(Hg.inv(self._frames[0].pose), self._frames[1].pose)

    def update(self, dt):
        self._pos0 = self._h01()[0:3, 3]

    def is_active(self):
        return True

    @property
    def jacobian(self):
        return (dot(Hg.adjoint(self._h01())[3:6, :], self._frames[1].jacobian)
                - self._frames[0].jacobian[3:6, :])

    def solve(self, vel, admittance, dt):
        dforce = -dot(pinv(admittance), vel + self._pos0 / dt)
        self._force += dforce
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
            (shapes, collision_solver) = choose_solver(shapes[0], shapes[1])
        self._shapes = shapes
        self._is_active = None
        self._sdist = None
        self._frames = (MovingSubFrame(shapes[0].frame.body),
                        MovingSubFrame(shapes[1].frame.body))
        self._collision_solver = collision_solver
        self._proximity = proximity

    def init(self, world):
        pass

    def update(self, dt):
        (sdist, H_gc0, H_gc1) = self._collision_solver(self._shapes)
        for k, H_gc in enumerate((H_gc0, H_gc1)):
            self._frames[k].bpose = dot(Hg.inv(self._shapes[k].frame.body.pose), H_gc)
        H_c0c1 = dot(Hg.inv(H_gc0), H_gc1)
        gap_rate = (dot(Hg.adjoint(H_c0c1)[5, :], self._frames[1].twist)
                    - self._frames[0].twist[5])
        self._is_active = (sdist + gap_rate * dt < self._proximity)
        self._sdist = sdist
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

    @property
    def ndol(self):
        return 4

    @property
    def jacobian(self):
        H_01 = dot(Hg.inv(self._frames[0].pose), self._frames[1].pose)
        return (dot(Hg.adjoint(H_01)[2:6, :], self._frames[1].jacobian)
                - self._frames[0].jacobian[2:6, :])

    def solve(self, vel, admittance, dt):
        free_vel = vel - dot(admittance, self._force)
        if self._sdist + dt * free_vel[3] > 0:
            released = -self._force
            self._force[:] = 0.
            return released
        # static friction: no relative motion at the contact
        target = hstack((vel[0:3], vel[3] + self._sdist / dt))
        dforce = dot(-pinv(admittance), target)
        candidate = self._force + dforce
        if sum((candidate[0:3] / self._eps) ** 2) <= (candidate[3] * self._mu) ** 2:
            self._force = candidate
            return dforce
        # sliding friction
        alpha = free_vel.copy()
        alpha[3] += self._sdist / dt
        y_col = admittance[0:3, 3]
        y_n = admittance[3, 3]
        beta = alpha[0:3] - alpha[3] / y_n * y_col
        a = self._mu / y_n * alpha[3]
        b = self._mu / y_n * y_col
        E = diag(self._eps ** 2)
        y_hat = admittance[0:3, 0:3] - dot(y_col, y_col) / y_n
        B = zeros((6, 6))
        B[0:3, 0:3] = dot(E, y_hat + 2 / a * dot(beta, b))
        B[0:3, 3:6] = -E * (dot(beta, beta) / a ** 2)
        B[3:6, 0:3] = E * dot(b, b) - eye(3)
        B[3:6, 3:6] = dot(E, y_hat)
        roots = eigvals(B)
        roots = roots[np.logical_and(roots.imag == 0, roots.real <= 0)]
        s = -1e10 if len(roots) == 0 else max(float(min(roots.real)), -1e10)
        previous = self._force.copy()
        A = admittance.copy()
        A[0:3, 0:3] -= s * diag(self._eps ** -2.)
        self._force = solve(A, -alpha)
        return self._force - previous


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
