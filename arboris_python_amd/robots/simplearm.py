"""Planar 3R arm (Shoulder / Elbow / Wrist hinges about z).

Same model as the reference's arboris/robots/simplearm.py:20-77: three box
links of the given lengths and masses, each with its mass matrix transported
from the centre to the proximal end, chained along +y.
"""
import numpy as np

from ..core import World, Body, SubFrame
from .. import homogeneousmatrix as Hg
from .. import massmatrix
from .. import shapes as _shapes
from ..joints import RzJoint

_LINKS = (("Arm", "Shoulder"), ("Forearm", "Elbow"), ("Hand", "Wrist"))
_ANCHORS = ("ElbowBaseFrame", "WristBaseFrame", "EndEffector")


def add_simplearm(world, name='', lengths=(0.5, 0.4, 0.2),
                  masses=(1.0, 0.8, 0.2), with_shapes=False):
    """Add the arm to ``world`` and call ``world.init()``."""
    assert isinstance(world, World)
    anchor = world.ground
    for (body_name, joint_name), anchor_name, length, mass in zip(
            _LINKS, _ANCHORS, lengths, masses):
        half_extents = (length / 20., length / 2., length / 20.)
        at_base = massmatrix.transport(massmatrix.box(half_extents, mass),
                                       Hg.transl(0., -length / 2., 0.))
        body = Body(name + body_name, at_base)
        if with_shapes:
            centre = SubFrame(body, Hg.transl(0., length / 2., 0.))
            world.register(_shapes.Box(centre, half_extents))
        world.add_link(anchor, RzJoint(name=name + joint_name), body)
        anchor = SubFrame(body, Hg.transl(0, length, 0), name + anchor_name)
    world.register(anchor)          # the end effector carries no joint
    world.init()
