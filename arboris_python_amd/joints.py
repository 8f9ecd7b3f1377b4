"""Built-in joint types (host-side definitions of the plugin API).

API mirror of arboris/joints.py: FreeJoint (:10-57), RzRyRxJoint (:59-104),
RzRyJoint (:107-146), RzRxJoint (:149-185), RyRxJoint (:188-224), RzJoint
(:227-303), RyJoint (:305-326), RxJoint (:328-349), TxTyTzJoint (:352-384).

Each class gives ``pose`` (H_rn, 4x4), ``jacobian`` (6 x ndof, rows [w; v]),
``djacobian`` and ``integrate``.  These NumPy versions serve model building,
inspection and user code; during stepping the same closed forms are evaluated on
the GPU by ``joint_local`` in csrc/arb_kernels.hip, selected by the type ids of
``flatten.JOINT_TYPES``.
"""
from numpy import array, zeros, eye, sin, cos, dot

from . import homogeneousmatrix as _Hg
from .core import Joint, LinearConfigurationSpaceJoint


def _cols(ndof, entries):
    """6 x ndof matrix from {(row, col): value}."""
    J = zeros((6, ndof))
    for (r, c), v in entries.items():
        J[r, c] = v
    return J


class FreeJoint(Joint):
    """6-dof joint; ``gpos`` is the 4x4 pose, ``gvel`` the body twist."""

    def __init__(self, gpos=None, gvel=None, name=None):
        self.gpos = eye(4) if gpos is None else array(gpos, dtype=float).reshape((4, 4))
        self.gvel = zeros(6) if gvel is None else array(gvel, dtype=float).reshape((6,))
        Joint.__init__(self, name)

    @property
    def ndof(self):
        return 6

    @property
    def pose(self):
        return self.gpos.copy()

    @property
    def twist(self):
        return self.gvel.copy()

    @property
    def jacobian(self):
        return eye(6)

    @property
    def djacobian(self):
        return zeros((6, 6))

    def integrate(self, gvel, dt):
        from .twistvector import exp
        self.gvel = gvel
        self.gpos = dot(self.gpos, exp(dt * self.gvel))


class RzRyRxJoint(LinearConfigurationSpaceJoint):
    """Ball joint as three serial hinges, H = Rz Ry Rx, gpos = (az, ay, ax)."""

    @property
    def ndof(self):
        return 3

    @property
    def pose(self):
        return _Hg.rotzyx(self.gpos[0], self.gpos[1], self.gpos[2])

    @property
    def jacobian(self):
        sy, cy = sin(self.gpos[1]), cos(self.gpos[1])
        sx, cx = sin(self.gpos[2]), cos(self.gpos[2])
        return _cols(3, {(0, 0): -sy, (1, 0): sx * cy, (2, 0): cx * cy,
                         (1, 1): cx, (2, 1): -sx,
                         (0, 2): 1.})

    @property
    def djacobian(self):
        sy, cy = sin(self.gpos[1]), cos(self.gpos[1])
        sx, cx = sin(self.gpos[2]), cos(self.gpos[2])
        dy, dx = self.gvel[1], self.gvel[2]
        return _cols(3, {(0, 0): -dy * cy,
                         (1, 0): dx * cx * cy - dy * sx * sy,
                         (2, 0): -dx * sx * cy - dy * cx * sy,
                         (1, 1): -dx * sx, (2, 1): -dx * cx})


class RzRyJoint(LinearConfigurationSpaceJoint):
    """Two serial hinges, H = Rz Ry, gpos = (az, ay)."""

    @property
    def ndof(self):
        return 2

    @property
    def pose(self):
        return _Hg.rotzy(self.gpos[0], self.gpos[1])

    @property
    def jacobian(self):
        sy, cy = sin(self.gpos[1]), cos(self.gpos[1])
        return _cols(2, {(0, 0): -sy, (2, 0): cy, (1, 1): 1.})

    @property
    def djacobian(self):
        sy, cy = sin(self.gpos[1]), cos(self.gpos[1])
        dy = self.gvel[1]
        return _cols(2, {(0, 0): -dy * cy, (2, 0): -dy * sy})


class RzRxJoint(LinearConfigurationSpaceJoint):
    """Two serial hinges, H = Rz Rx, gpos = (az, ax)."""

    @property
    def ndof(self):
        return 2

    @property
    def pose(self):
        return _Hg.rotzx(self.gpos[0], self.gpos[1])

    @property
    def jacobian(self):
        sx, cx = sin(self.gpos[1]), cos(self.gpos[1])
        return _cols(2, {(1, 0): sx, (2, 0): cx, (0, 1): 1.})

    @property
    def djacobian(self):
        sx, cx = sin(self.gpos[1]), cos(self.gpos[1])
        dx = self.gvel[1]
        return _cols(2, {(1, 0): dx * cx, (2, 0): -dx * sx})


class RyRxJoint(LinearConfigurationSpaceJoint):
    """Two serial hinges, H = Ry Rx, gpos = (ay, ax)."""

    @property
    def ndof(self):
        return 2

    @property
    def pose(self):
        return _Hg.rotyx(self.gpos[0], self.gpos[1])

    @property
    def jacobian(self):
        sx, cx = sin(self.gpos[1]), cos(self.gpos[1])
        return _cols(2, {(1, 0): cx, (2, 0): -sx, (0, 1): 1.})

    @property
    def djacobian(self):
        sx, cx = sin(self.gpos[1]), cos(self.gpos[1])
        dx = self.gvel[1]
        return _cols(2, {(1, 0): -dx * sx, (2, 0): -dx * cx})


class _Hinge(LinearConfigurationSpaceJoint):
    """1-dof hinge about a coordinate axis of the joint frame."""
    _axis = None          # row of the unit angular velocity
    _rot = None           # homogeneousmatrix.rot{x,y,z}

    @property
    def ndof(self):
        return 1

    @property
    def pose(self):
        return type(self)._rot(self.gpos[0])

    @property
    def ipose(self):
        return type(self)._rot(-self.gpos[0])

    @property
    def jacobian(self):
        return _cols(1, {(self._axis, 0): 1.})

    @property
    def djacobian(self):
        return zeros((6, 1))


class RzJoint(_Hinge):
    """Hinge about z."""
    _axis = 2
    _rot = staticmethod(_Hg.rotz)


class RyJoint(_Hinge):
    """Hinge about y."""
    _axis = 1
    _rot = staticmethod(_Hg.roty)


class RxJoint(_Hinge):
    """Hinge about x."""
    _axis = 0
    _rot = staticmethod(_Hg.rotx)


class TxTyTzJoint(LinearConfigurationSpaceJoint):
    """Three serial prismatic joints, H = transl(gpos)."""

    @property
    def ndof(self):
        return 3

    @property
    def pose(self):
        return _Hg.transl(self.gpos[0], self.gpos[1], self.gpos[2])

    @property
    def jacobian(self):
        return _cols(3, {(3, 0): 1., (4, 1): 1., (5, 2): 1.})

    @property
    def djacobian(self):
        return zeros((6, 3))
