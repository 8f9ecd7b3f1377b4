"""Narrow-phase collision solvers between shape pairs (host-side plugin API).

API mirror of arboris/collisions.py: ``choose_solver`` (:14-65) and the
closed-form pair solvers (:67-299).  Every solver takes the ordered pair of
shapes and returns ``(sdist, H_gc0, H_gc1)``: the signed distance and the two
contact frames (z axis = contact normal, pointing from shape 0 to shape 1).

During stepping the plane-point / plane-sphere case is evaluated on the GPU
(``contact_update`` in csrc/arb_kernels.hip); these NumPy versions serve
``choose_solver``-based contact discovery and user code.
"""
import numpy as np
from numpy.linalg import norm

from . import homogeneousmatrix as Hg
from .core import Shape
from .shapes import Plane, Point, Box, Cylinder, Sphere

_ORDER = {}     # (type0, type1) -> (swap?, solver name); filled below


def choose_solver(shape0, shape1):
    """Return ``((shape_a, shape_b), solver)`` for the pair, possibly swapped
    so that the solver's expected ordering holds.  Raises NotImplementedError
    for unsupported pairs."""
    assert isinstance(shape0, Shape)
    assert isinstance(shape1, Shape)
    for (t0, t1), (swap, solver) in _ORDER.items():
        if isinstance(shape0, t0) and isinstance(shape1, t1):
            return (((shape1, shape0) if swap else (shape0, shape1)), solver)
    raise NotImplementedError()


def _origin(shape):
    return shape.frame.pose[0:3, 3]


def sphere_sphere_collision(shapes):
    assert isinstance(shapes[0], Sphere) and isinstance(shapes[1], Sphere)
    return _sphere_sphere_collision(_origin(shapes[0]), shapes[0].radius,
                                    _origin(shapes[1]), shapes[1].radius)


def sphere_point_collision(shapes):
    assert isinstance(shapes[0], Sphere) and isinstance(shapes[1], Point)
    return _sphere_sphere_collision(_origin(shapes[0]), shapes[0].radius,
                                    _origin(shapes[1]), 0.)


def box_sphere_collision(shapes):
    assert isinstance(shapes[0], Box) and isinstance(shapes[1], Sphere)
    return _box_sphere_collision(shapes[0].frame.pose, shapes[0].half_extents,
                                 _origin(shapes[1]), shapes[1].radius)


def box_point_collision(shapes):
    assert isinstance(shapes[0], Box) and isinstance(shapes[1], Point)
    return _box_sphere_collision(shapes[0].frame.pose, shapes[0].half_extents,
                                 _origin(shapes[1]), 0.)


def plane_sphere_collision(shapes):
    assert isinstance(shapes[0], Plane) and isinstance(shapes[1], Sphere)
    return _plane_sphere_collision(shapes[0].frame.pose, shapes[0].coeffs,
                                   _origin(shapes[1]), shapes[1].radius)


def plane_point_collision(shapes):
    assert isinstance(shapes[0], Plane) and isinstance(shapes[1], Point)
    return _plane_sphere_collision(shapes[0].frame.pose, shapes[0].coeffs,
                                   _origin(shapes[1]), 0.)


def _frames_along(normal, origin0, origin1):
    H0 = Hg.zaligned(normal)
    H1 = H0.copy()
    H0[0:3, 3] = origin0
    H1[0:3, 3] = origin1
    return H0, H1


def _sphere_sphere_collision(p_g0, radius0, p_g1, radius1):
    """Two spheres given by centre and radius (a point is a 0-radius sphere)."""
    delta = np.asarray(p_g1, float) - np.asarray(p_g0, float)
    dist = norm(delta)
    sdist = dist - radius0 - radius1
    normal = delta / dist
    on0 = p_g0 + radius0 * normal
    H_gc0, H_gc1 = _frames_along(normal, on0, on0 + sdist * normal)
    return (sdist, H_gc0, H_gc1)


def _plane_sphere_collision(H_g0, coeffs0, p_g1, radius1):
    """Plane (pose ``H_g0``, coefficients (n, d)) against a sphere.

    As in the reference, the contact frames are expressed with the sphere
    centre taken in the plane's coordinates.
    """
    assert Hg.ishomogeneousmatrix(H_g0)
    assert norm(coeffs0[0:3]) == 1.
    assert radius1 >= 0.
    normal = coeffs0[0:3]
    centre = Hg.pdot(Hg.inv(H_g0), p_g1)
    centre_dist = np.dot(normal, centre) - coeffs0[3]
    sdist = centre_dist - radius1
    H_gc0, H_gc1 = _frames_along(normal, centre - centre_dist * normal,
                                 centre - np.sign(sdist) * radius1 * normal)
    return (sdist, H_gc0, H_gc1)


def _box_sphere_collision(H_g0, half_extents0, p_g1, radius1):
    """Box (pose, half extents) against a sphere."""
    assert Hg.ishomogeneousmatrix(H_g0)
    half = np.asarray(half_extents0, float)
    local = Hg.pdot(Hg.inv(H_g0), p_g1)
    if (abs(local) <= half).all():
        # centre inside the box: push out through the nearest face
        gaps = np.hstack((half - local, half + local))
        i = int(np.argmin(gaps))
        face = local.copy()
        normal = np.zeros(3)
        if i < 3:
            face[i] = half[i]
            normal[i] = 1
        else:
            face[i - 3] = -half[i - 3]
            normal[i - 3] = -1
        f_g = Hg.pdot(H_g0, face)
        sdist = -norm(f_g - p_g1) - radius1
    else:
        nearest = np.clip(local, -half, half)
        f_g = Hg.pdot(H_g0, nearest)
        delta = p_g1 - f_g
        normal = delta / norm(delta)
        sdist = norm(delta) - radius1
    H_gc0, H_gc1 = _frames_along(normal, f_g, p_g1 - radius1 * normal)
    return (sdist, H_gc0, H_gc1)


_ORDER.update({
    (Sphere, Sphere): (False, sphere_sphere_collision),
    (Sphere, Point): (False, sphere_point_collision),
    (Sphere, Plane): (True, plane_sphere_collision),
    (Sphere, Box): (True, box_sphere_collision),
    (Point, Sphere): (True, sphere_point_collision),
    (Point, Plane): (True, plane_point_collision),
    (Plane, Sphere): (False, plane_sphere_collision),
    (Plane, Point): (False, plane_point_collision),
    (Box, Sphere): (False, box_sphere_collision),
    (Box, Point): (False, box_point_collision),
})
