"""Star-import convenience, mirroring arboris/all.py:13-24."""
from . import core, homogeneousmatrix, twistvector, adjointmatrix, massmatrix
from . import joints, shapes, collisions, constraints, controllers
from .core import (World, Body, Frame, SubFrame, MovingSubFrame, Joint, JointsList,
                   LinearConfigurationSpaceJoint, NamedObjectsList, Constraint,
                   Controller, Shape, Observer, simulate)
from .joints import *
from .shapes import *
from .controllers import *
from .constraints import *
from .robots.simplearm import add_simplearm
from .robots.snake import add_snake
from .robots.human36 import add_human36
from .robots.simpleshapes import add_sphere, add_box, add_cylinder, add_groundplane
from .visu_collada import write_collada_animation, write_collada_scene
from . import observers
from .observers import EnergyMonitor, PerfMonitor, Hdf5Logger
from numpy import arange, dot, allclose, pi
