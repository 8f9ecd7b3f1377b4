"""The worlds of BASELINE.json's configurations, built with this package's API.

Each builder returns an initialised ``core.World``; ``flat(world)`` lowers it to
the batch-shared ``FlatModel``.  The contact subsets follow SURVEY.md section 8d:
the reference's falling-human test registers the 8 foot points
(tests/test_human36_falling.py:30-37); BASELINE's "4 floor contacts" is the
heel + toe-tip subset.
"""
import numpy as np

from .core import World
from .flatten import flatten_world
from .controllers import WeightController
from .constraints import get_all_contacts
from .robots.simplearm import add_simplearm
from .robots.snake import add_snake
from .robots.human36 import add_human36
from .robots.simpleshapes import add_groundplane

FOUR_CONTACTS = ('Right foot toe tip', 'Right foot heel',
                 'Left foot toe tip', 'Left foot heel')


def simplearm_world(gravity=True):
    w = World()
    if gravity:
        w.register(WeightController())
    add_simplearm(w)
    return w


def snake_world(nbody=64, gravity=True, is_fixed=True, **kw):
    w = World()
    add_snake(w, nbody, is_fixed=is_fixed, **kw)
    if gravity:
        w.register(WeightController())
        w.init()
    return w


def human36_world(contacts=0, gravity=True, friction_coeff=0.6):
    """human36 (+ ground plane and 4 or 8 SoftFingerContacts)."""
    assert contacts in (0, 4, 8)
    w = World()
    if contacts:
        add_groundplane(w)
    add_human36(w)
    if gravity:
        w.register(WeightController())
    if contacts:
        for c in get_all_contacts(w, friction_coeff=friction_coeff):
            if contacts == 8 or c._shapes[1].name in FOUR_CONTACTS:
                w.register(c)
    w.init()
    return w


def flat(world):
    """FlatModel of an initialised world."""
    return flatten_world(world)[0]
