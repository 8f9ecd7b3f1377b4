"""SE(3) helpers on 4x4 homogeneous matrices (host side, model-build time).

API mirror of the reference module arboris/homogeneousmatrix.py (transl :11-31,
rotzyx :34-59, rotzy :60-80, rotzx :82-102, rotyx :104-124, rotx/roty/rotz
:126-199, zaligned :201-232, ishomogeneousmatrix :234-240, pdot/vdot :242-252,
inv :254-275, adjoint :277-319, iadjoint :321-325, rotzyx_angles :327-353).
Conventions kept: twists are ordered [angular; linear], Ad(H) = [[R,0],[p^R,R]],
composite rotations are R = Rz.Ry.Rx.

These functions run on the host with NumPy; the per-step device code has its own
implementation (csrc/arb_kernels.hip).
"""
import numpy as np
from numpy import sin, cos, arctan2

tol = 1e-9


def _hom(R, p=(0., 0., 0.)):
    H = np.eye(4)
    H[0:3, 0:3] = R
    H[0:3, 3] = p
    return H


def transl(t_x, t_y, t_z):
    """Pure translation."""
    return _hom(np.eye(3), (t_x, t_y, t_z))


def rotx(angle):
    c, s = cos(angle), sin(angle)
    return _hom([[1., 0., 0.], [0., c, -s], [0., s, c]])


def roty(angle):
    c, s = cos(angle), sin(angle)
    return _hom([[c, 0., s], [0., 1., 0.], [-s, 0., c]])


def rotz(angle):
    c, s = cos(angle), sin(angle)
    return _hom([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]])


def rotzyx(angle_z, angle_y, angle_x):
    """R = Rz(angle_z) Ry(angle_y) Rx(angle_x) (closed form)."""
    sz, cz = sin(angle_z), cos(angle_z)
    sy, cy = sin(angle_y), cos(angle_y)
    sx, cx = sin(angle_x), cos(angle_x)
    return _hom([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                 [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                 [-sy, cy * sx, cy * cx]])


def rotzy(angle_z, angle_y):
    """R = Rz Ry."""
    sz, cz = sin(angle_z), cos(angle_z)
    sy, cy = sin(angle_y), cos(angle_y)
    return _hom([[cz * cy, -sz, cz * sy],
                 [sz * cy, cz, sz * sy],
                 [-sy, 0., cy]])


def rotzx(angle_z, angle_x):
    """R = Rz Rx."""
    sz, cz = sin(angle_z), cos(angle_z)
    sx, cx = sin(angle_x), cos(angle_x)
    return _hom([[cz, -sz * cx, sz * sx],
                 [sz, cz * cx, -cz * sx],
                 [0., sx, cx]])


def rotyx(angle_y, angle_x):
    """R = Ry Rx."""
    sy, cy = sin(angle_y), cos(angle_y)
    sx, cx = sin(angle_x), cos(angle_x)
    return _hom([[cy, sy * sx, sy * cx],
                 [0., cx, -sx],
                 [-sy, cy * sx, cy * cx]])


def zaligned(vec):
    """Frame whose z axis is the unit vector ``vec``.

    The x axis is built from the two largest-magnitude components of ``vec``
    (same tie-breaking as the reference: ``argsort`` of |vec|).
    """
    z = np.array(vec, dtype=float).reshape(3)
    assert abs(np.linalg.norm(z) - 1) < 1e-9
    order = np.argsort(np.absolute(z))
    x = np.zeros(3)
    x[order[1]] = z[order[2]]
    x[order[2]] = -z[order[1]]
    x /= np.linalg.norm(x)
    H = np.eye(4)
    H[0:3, 0] = x
    H[0:3, 1] = np.cross(z, x)
    H[0:3, 2] = z
    return H


def ishomogeneousmatrix(H, tol=tol):
    """True for a 4x4 matrix with det(R) = 1 (to ``tol``) and last row 0 0 0 1."""
    H = np.asarray(H)
    return (H.shape == (4, 4)
            and bool(abs(np.linalg.det(H[0:3, 0:3]) - 1) <= tol)
            and bool((H[3, 0:4] == [0, 0, 0, 1]).all()))


def pdot(H, point):
    """Change of frame for a point."""
    assert ishomogeneousmatrix(H)
    return np.dot(H[0:3, 0:3], point) + H[0:3, 3]


def vdot(H, vec):
    """Change of frame for a free vector."""
    assert ishomogeneousmatrix(H)
    return np.dot(H[0:3, 0:3], vec)


def inv(H):
    """Inverse of a homogeneous matrix: [[R^T, -R^T p], [0, 1]]."""
    assert ishomogeneousmatrix(H)
    Rt = H[0:3, 0:3].T
    return _hom(Rt, -np.dot(Rt, H[0:3, 3]))


def adjoint(H):
    """6x6 adjoint Ad(H) = [[R, 0], [p^ R, R]]."""
    assert ishomogeneousmatrix(H), H
    R = H[0:3, 0:3]
    px, py, pz = H[0:3, 3]
    phat = np.array([[0., -pz, py], [pz, 0., -px], [-py, px, 0.]])
    Ad = np.zeros((6, 6))
    Ad[0:3, 0:3] = R
    Ad[3:6, 3:6] = R
    Ad[3:6, 0:3] = np.dot(phat, R)
    return Ad


def iadjoint(H):
    """Adjoint of the inverse matrix."""
    return adjoint(inv(H))


def rotzyx_angles(H):
    """Angles (az, ay, ax) such that R = Rz(az) Ry(ay) Rx(ax)."""
    assert ishomogeneousmatrix(H)
    if abs(H[0, 0]) < tol and abs(H[1, 0]) < tol:
        az = 0
        ay = arctan2(-H[2, 0], H[0, 0])
        ax = arctan2(-H[1, 2], H[1, 1])
    else:
        az = arctan2(H[1, 0], H[0, 0])
        sz, cz = sin(az), cos(az)
        ay = arctan2(-H[2, 0], cz * H[0, 0] + sz * H[1, 0])
        ax = arctan2(sz * H[0, 2] - cz * H[1, 2], cz * H[1, 1] - sz * H[0, 1])
    return (az, ay, ax)
