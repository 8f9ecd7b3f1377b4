"""6x6 body mass matrices (host side, model-build time).

API mirror of arboris/massmatrix.py: ``ismassmatrix`` (:11-26), ``transport``
(:28-69), ``principalframe`` (:72-107) and the homogeneous solids ``box``,
``ellipsoid``, ``cylinder``, ``sphere`` (:109-180).  Layout: rows/columns
ordered [angular; linear], i.e. M = [[I, m c^], [m c^T, m 1]].
"""
import numpy as np

from . import homogeneousmatrix as Hg


def ismassmatrix(M, semi=False):
    """True when M is 6x6, symmetric, has an isotropic linear block and is
    positive (semi-)definite."""
    M = np.asarray(M)
    if M.shape != (6, 6) or not np.allclose(M, M.T):
        return False
    if not np.allclose(M[3:6, 3:6], M[3, 3] * np.eye(3)):
        return False
    ev = np.linalg.eigvals(M)
    return bool((ev >= 0.).all()) if semi else bool((ev > 0.).all())


def transport(M, H):
    """Express the mass matrix in another frame: Ad(H)^T M Ad(H), H = H_ab."""
    assert ismassmatrix(M)
    assert Hg.ishomogeneousmatrix(H)
    Ad = Hg.adjoint(H)
    return np.dot(Ad.T, np.dot(M, Ad))


def principalframe(M):
    """Homogeneous matrix from the frame of M to its principal inertia frame."""
    assert ismassmatrix(M)
    m = M[5, 5]
    rx = M[0:3, 3:6] / m
    H = np.eye(4)
    H[0:3, 3] = [rx[2, 1], rx[0, 2], rx[1, 0]]
    central = M[0:3, 0:3] + m * np.dot(rx, rx)
    (S, R) = np.linalg.eig(central)
    if np.linalg.det(R) < 0.:
        flip = np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]])
        R = np.dot(R, flip)
    H[0:3, 0:3] = R
    return H


def _solid(Ix, Iy, Iz, mass):
    return np.diag((Ix, Iy, Iz, mass, mass, mass)).astype(float)


def box(half_extents, mass):
    """Homogeneous parallelepiped, at its centre."""
    (x, y, z) = half_extents
    k = mass / 3.
    return _solid(k * (y ** 2 + z ** 2), k * (x ** 2 + z ** 2), k * (x ** 2 + y ** 2), mass)


def ellipsoid(radii, mass):
    """Homogeneous ellipsoid, at its centre."""
    (x, y, z) = radii
    k = mass / 5.
    return _solid(k * (y ** 2 + z ** 2), k * (x ** 2 + z ** 2), k * (x ** 2 + y ** 2), mass)


def cylinder(length, radius, mass):
    """Homogeneous cylinder whose axis is z, at its centre."""
    tangent = mass * (radius ** 2 / 4. + length ** 2 / 12.)
    axial = mass * radius ** 2 / 2.
    return _solid(tangent, tangent, axial, mass)


def sphere(radius, mass):
    """Homogeneous sphere, at its centre."""
    i = 2. * mass * radius ** 2 / 5.
    return _solid(i, i, i, mass)
