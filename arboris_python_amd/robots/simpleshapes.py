"""Worlds made of one free-floating shape, and the ground plane.

Same builders as the reference's arboris/robots/simpleshapes.py:12-76.
"""
from ..joints import FreeJoint
from ..shapes import Sphere, Box, Cylinder, Plane
from ..core import World, Body
from .. import massmatrix


def _add_free_body(world, body, shape):
    assert isinstance(world, World)
    world.add_link(world.ground, FreeJoint(), body)
    world.register(shape)
    world.init()


def add_sphere(world, radius=1., mass=1., name=None):
    ball = Body(name=name, mass=massmatrix.sphere(radius, mass))
    _add_free_body(world, ball, Sphere(ball, radius))


def add_box(world, half_extents=(1., 1., 1.), mass=1., name='Box'):
    body = Body(name=name, mass=massmatrix.box(half_extents, mass))
    _add_free_body(world, body, Box(body, half_extents))


def add_cylinder(world, length=1., radius=1., mass=1., name='Cylinder'):
    body = Body(name=name, mass=massmatrix.cylinder(length, radius, mass))
    _add_free_body(world, body, Cylinder(body, length, radius))


def add_groundplane(world):
    """Plane through the ground origin whose normal is ``world.up``."""
    coeffs = list(world.up) + [0.]
    world.register(Plane(world.ground, coeffs, 'Ground shape'))
