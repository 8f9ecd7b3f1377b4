"""Helpers on 6x6 adjoint matrices Ad(H) = [[R, 0], [p^R, R]] (host side).

API mirror of arboris/adjointmatrix.py (``isadjointmatrix`` :13-21, ``inv``
:24-32).  Nothing on the simulation path calls these; they are kept so that user
code importing them keeps working.
"""
import numpy as np


def isadjointmatrix(a):
    """Loose structural check of an adjoint matrix."""
    a = np.asarray(a)
    return (a.shape == (6, 6)
            and bool(np.linalg.det(a[0:3, 0:3]) == 1)
            and bool((a[0:3, 0:3] == a[3:6, 3:6]).all())
            and bool((a[0:3, 3:6] == 0).all()))


def inv(Ad):
    """Inverse of an adjoint matrix: both 3x3 blocks are transposed."""
    Ad = np.asarray(Ad)
    out = np.zeros((6, 6))
    out[0:3, 0:3] = Ad[0:3, 0:3].T
    out[3:6, 3:6] = Ad[0:3, 0:3].T
    out[3:6, 0:3] = Ad[3:6, 0:3].T
    return out
