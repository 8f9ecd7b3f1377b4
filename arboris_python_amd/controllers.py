"""Controllers: generalized forces + first-order impedance (host-side plugin API).

API mirror of arboris/controllers.py: WeightController (:10-60) and
ProportionalDerivativeController (:63-158).  ``update(dt)`` returns
``(gforce, impedance)`` as in the reference.  During stepping both controllers
are lowered to constants by ``flatten.flatten_world`` (gravity vector; merged
Kp/Kd matrices and the constant torque) and applied inside the HIP kernel
(``body_pass``/``pd_columns`` in csrc/arb_kernels.hip); the NumPy ``update``
methods below exist for inspection and for user code built on the API.
"""
from numpy import array, zeros, dot, ix_
from numpy.linalg import norm

from . import homogeneousmatrix as _Hg
from .core import Controller, World
from .joints import LinearConfigurationSpaceJoint


class WeightController(Controller):
    """Gravity, applied to every body with a non-zero mass matrix.

    gforce = sum_b J_b^T M_b Ad(H_gb^-1) [0; g * up]; zero impedance.
    """

    def __init__(self, gravity=-9.81, name=None):
        self.gravity = float(gravity)
        Controller.__init__(self, name=name)
        self._bodies = None
        self._wndof = None
        self._gravity_dtwist = None

    def init(self, world):
        assert isinstance(world, World)
        self._bodies = [b for b in world.ground.iter_descendant_bodies()
                        if norm(b.mass > 0.)]
        self._wndof = world.ndof
        self._gravity_dtwist = zeros(6)
        self._gravity_dtwist[3:6] = self.gravity * world.up

    def update(self, dt=None):
        gforce = zeros(self._wndof)
        for b in self._bodies:
            g_body = dot(_Hg.iadjoint(b.pose), self._gravity_dtwist)
            gforce += dot(b.jacobian.T, dot(b.mass, g_body))
        return (gforce, zeros((self._wndof, self._wndof)))


class ProportionalDerivativeController(Controller):
    """tau = Kp (q_d - q(t+dt)) + Kd (dq_d - dq(t+dt)) on a set of joints.

    With q(t+dt) = q(t) + dt dq(t+dt) this splits into the constant part
    tau_0 = Kp (q_d - q) + Kd dq_d and the impedance Z = -(dt Kp + Kd).
    """

    def __init__(self, joints, kp=None, kd=None, gpos_des=None, gvel_des=None,
                 name=None):
        Controller.__init__(self, name=name)
        joints = list(joints)
        n = 0
        dof_map = []
        for j in joints:
            if not isinstance(j, LinearConfigurationSpaceJoint):
                raise ValueError('Joints must be LinearConfigurationSpaceJoint instances')
            n += j.ndof
            dof_map.extend(range(j.dof.start, j.dof.stop))
        self._cndof = n
        self._dof_map = array(dof_map)
        self.joints = joints
        self._wndof = None

        def _mat(x):
            return zeros((n, n)) if x is None else array(x, dtype=float).reshape((n, n))

        def _vec(x):
            return zeros(n) if x is None else array(x, dtype=float).reshape(n)
        self.kp = _mat(kp)
        self.kd = _mat(kd)
        self.gpos_des = _vec(gpos_des)
        self.gvel_des = _vec(gvel_des)

    def init(self, world):
        self._wndof = world.ndof
        dof_map = []
        for j in self.joints:
            dof_map.extend(range(j.dof.start, j.dof.stop))
        self._dof_map = array(dof_map)

    def update(self, dt):
        gpos = array([x for j in self.joints for x in j.gpos]).reshape(self._cndof)
        gforce = zeros(self._wndof)
        impedance = zeros((self._wndof, self._wndof))
        gforce[self._dof_map] = (dot(self.kp, self.gpos_des - gpos)
                                 + dot(self.kd, self.gvel_des))
        impedance[ix_(self._dof_map, self._dof_map)] = -(dt * self.kp + self.kd)
        return (gforce, impedance)
