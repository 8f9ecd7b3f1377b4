"""Collision shapes (host-side plugin API).

API mirror of arboris/shapes.py: Plane (:9-30), Point (:32-38), Box (:40-47),
Cylinder (:49-58), Sphere (:60-68).
"""
import numpy

from .core import Shape


class Plane(Shape):
    """Plane a x + b y + c z + d = 0 in the coordinates of ``frame``; the
    normal (a, b, c) is normalised at construction."""

    def __init__(self, frame, coeffs=(0., 1., 0., 0.), name=None):
        Shape.__init__(self, frame, name)
        coeffs = numpy.array(coeffs, dtype=float)
        self.coeffs = coeffs / numpy.linalg.norm(coeffs[0:3])


class Point(Shape):
    """The origin of ``frame``."""

    def __init__(self, frame, name=None):
        Shape.__init__(self, frame, name)


class Box(Shape):
    """Axis-aligned box centred on ``frame``."""

    def __init__(self, frame, half_extents=(1., 1., 1.), name=None):
        Shape.__init__(self, frame, name)
        self.half_extents = half_extents


class Cylinder(Shape):
    """Cylinder whose symmetry axis is the z axis of ``frame``."""

    def __init__(self, frame, length=1., radius=1., name=None):
        assert radius >= 0.
        Shape.__init__(self, frame, name)
        self.radius = radius
        self.length = length


class Sphere(Shape):
    """Sphere centred on ``frame``."""

    def __init__(self, frame, radius=1., name=None):
        assert radius >= 0.
        Shape.__init__(self, frame, name)
        self.radius = radius
