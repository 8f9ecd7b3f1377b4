"""Derived quantities of a rigid motion (host side mixin for joints).

API mirror of arboris/rigidmotion.py:10-73.  With ``n`` the moving ("new")
frame and ``r`` the reference frame: ``pose`` = H_rn, ``twist`` = T_nr, and the
properties below derive the inverse pose, the adjoints and their time
derivatives exactly as the reference defines them (including its sign and frame
conventions for ``itwist``).
"""
from abc import ABCMeta, abstractproperty

import numpy as np

from . import homogeneousmatrix as Hg
from . import twistvector as T


class RigidMotion(object, metaclass=ABCMeta):

    @abstractproperty
    def pose(self):
        """H_rn as a 4x4 homogeneous matrix."""

    @abstractproperty
    def twist(self):
        """T_nr as a (6,) twist."""

    @property
    def ipose(self):
        return Hg.inv(self.pose)

    @property
    def itwist(self):
        return -np.dot(self.iadjoint, self.twist)

    @property
    def adjoint(self):
        return Hg.adjoint(self.pose)

    @property
    def iadjoint(self):
        return Hg.adjoint(self.ipose)

    @property
    def adjacency(self):
        return T.adjacency(self.twist)

    @property
    def iadjacency(self):
        return T.adjacency(self.itwist)

    @property
    def dadjoint(self):
        return np.dot(self.adjoint, self.adjacency)

    @property
    def idadjoint(self):
        return np.dot(self.iadjoint, self.iadjacency)
