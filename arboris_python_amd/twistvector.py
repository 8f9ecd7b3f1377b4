"""se(3) helpers for twists stored as [angular; linear] (host side).

API mirror of arboris/twistvector.py: ``adjacency`` (:10-33) and ``exp`` (:35-70).
The device-side counterparts live in csrc/arb_kernels.hip.
"""
import numpy as np


def _hat(a):
    return np.array([[0., -a[2], a[1]],
                     [a[2], 0., -a[0]],
                     [-a[1], a[0], 0.]])


def adjacency(tw):
    """6x6 matrix ad(T) = [[w^, 0], [v^, w^]] of the twist ``tw`` = [w; v]."""
    tw = np.asarray(tw, dtype=float)
    assert tw.shape == (6,)
    ad = np.zeros((6, 6))
    what = _hat(tw[0:3])
    ad[0:3, 0:3] = what
    ad[3:6, 3:6] = what
    ad[3:6, 0:3] = _hat(tw[3:6])
    return ad


def exp(tw):
    """Exponential map se(3) -> SE(3) (Rodrigues; series below |w| = 1e-3)."""
    tw = np.asarray(tw, dtype=float)
    assert tw.shape == (6,)
    w, v = tw[0:3], tw[3:6]
    what = _hat(w)
    theta = np.linalg.norm(w)
    if theta >= 0.001:
        cc = (1 - np.cos(theta)) / theta ** 2
        sc = np.sin(theta) / theta
        dsc = (theta - np.sin(theta)) / theta ** 3
    else:
        cc = 1. / 2.
        sc = 1. - theta ** 2 / 6.
        dsc = 1. / 6.
    H = np.eye(4)
    H[0:3, 0:3] = np.eye(3) + sc * what + cc * np.dot(what, what)
    H[0:3, 3] = np.dot(sc * np.eye(3) + cc * what + dsc * np.outer(w, w), v)
    return H
