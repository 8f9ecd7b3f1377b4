"""Planar N-link chain of z hinges ("snake").

Same model as the reference's arboris/robots/snake.py:17-60: cylinders of
radius length/10 along +y, mass matrix transported to the proximal end; the base
is either fixed to the ground or carried by a FreeJoint cube.
"""
from numpy import dot

from ..core import World, Body, SubFrame
from ..massmatrix import transport, cylinder, box
from ..homogeneousmatrix import transl, rotx
from ..joints import FreeJoint, RzJoint


def add_snake(w, nbody, lengths=None, masses=None, gpos=None, gvel=None,
              is_fixed=True):
    """Add the chain to ``w`` and call ``w.init()``."""
    assert isinstance(w, World)
    lengths = [.5] * nbody if lengths is None else lengths
    masses = [2.] * nbody if masses is None else masses
    gpos = [0.] * nbody if gpos is None else gpos
    gvel = [0.] * nbody if gvel is None else gvel
    for seq in (lengths, masses, gpos, gvel):
        assert nbody == len(seq)
    if is_fixed:
        anchor = w.ground
    else:
        half = lengths[0] / 2.
        anchor = Body(mass=box([half, half, half], masses[0]))
        w.add_link(w.ground, FreeJoint(), anchor)
    for (length, mass, q, dq) in zip(lengths, masses, gpos, gvel):
        to_base = dot(rotx(0.), transl(0., -length / 2., 0.))
        link = Body(mass=transport(cylinder(length, length / 10., mass), to_base))
        w.add_link(anchor, RzJoint(gpos=q, gvel=dq), link)
        anchor = SubFrame(link, transl(0., length, 0.))
    w.register(anchor)
    w.init()
