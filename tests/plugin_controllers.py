"""A user-defined Controller, written ONCE against the reference's plugin API (core.py:327-339: ``init(world)``,
``update(dt) -> (gforce, impedance)``) and instantiated over either package's ``Controller`` base class:

* ``tools/gen_golden.py`` (g14) derives it from the REFERENCE's ``arboris.core.Controller`` and records what the
  reference's ``simulate`` loop does with it;
* ``tests/test_gpu_round6.py`` derives it from ``arboris_python_amd.core.Controller`` and runs the same loop on the
  device through the generic plugin path (``ext_gforce`` / ``ext_impedance``, ABI 8).

It only touches API both packages share: ``world.getjoints()``, ``joint.dof``, ``joint.gpos``, ``world.ndof``.
"""
import numpy as np


def make_spring_damper(controller_base):
    class JointSpringDamper(controller_base):
        """Joint-space spring and damper on every hinge dof plus a dense coupling term:

            tau = -K q(t+dt) - (D + c u u^T) dq(t+dt),   q(t+dt) = q(t) + dt dq(t+dt)

        i.e. gforce_0 = -K q(t), impedance = -(dt K + D + c u u^T) with u = 1 on the hinge dofs: a symmetric, DENSE
        impedance that couples every pair of hinge dofs (nothing the built-in controllers can express)."""

        def __init__(self, stiffness=8., damping=(0.5, 2.5), coupling=0.05, name=None):
            controller_base.__init__(self, name=name)
            self.stiffness, self.damping, self.coupling = float(stiffness), tuple(damping), float(coupling)
            self._world = None

        def init(self, world):
            self._world = world
            n = world.ndof
            self._hinge = [j for j in world.getjoints() if np.ndim(j.gpos) == 1]
            self._u = np.zeros(n)
            for j in self._hinge:
                self._u[j.dof] = 1.
            # a different damping on every dof: (lo .. hi) along the dof index
            lo, hi = self.damping
            self._d = self._u * np.linspace(lo, hi, n)

        def update(self, dt):
            n = self._world.ndof
            q = np.zeros(n)
            for j in self._hinge:
                q[j.dof] = j.gpos
            k = self.stiffness * self._u
            gforce = -k * q
            impedance = -(np.diag(dt * k + self._d) + self.coupling * np.outer(self._u, self._u))
            return (gforce, impedance)

    return JointSpringDamper
