"""World / Body / Joint / Constraint / Controller plugin API on top of the MI355X step.

This module is the host-side mirror of the reference's ``arboris/core.py``:
the same classes, method names, argument meaning and error behaviour, so that
robot definitions written against arboris (``robots/*.py``-style builders) load
unchanged.  What differs is *where the step runs*: the reference evaluates

    update_dynamic -> update_controllers -> update_constraints -> integrate
    (arboris/core.py:682-980, loop at :1356-1363)

in NumPy on the object graph; here those four methods flatten the tree
(``flatten.py``, counterpart of ``World.init`` core.py:608-635) and run the
hand-written HIP kernels of ``libarbstep.so`` through the ctypes C ABI
(``_capi.py`` / ``include/arbstep.h``) on a batch of one world, then scatter the
results back onto the objects (``Body.pose``, ``World.mass`` ...).  Large
batches of worlds use ``arboris_python_amd.batch.BatchedWorlds`` directly.

There is no CPU fallback: without the built library (or without a GPU) the
four step methods raise ``RuntimeError``.

Reference map: NamedObject/NamedObjectsList core.py:38-126, Frame :129-156,
Joint :158-220, LinearConfigurationSpaceJoint :223-240, JointsList :243-267,
Constraint :269-315, Shape :318-324, Controller :327-339, World :342-980,
SubFrame/MovingSubFrame :983-1053, Body :1055-1315, Observer :1318-1331,
simulate :1334-1365.
"""
from abc import ABCMeta, abstractmethod, abstractproperty

import numpy
from numpy import array, zeros, eye, dot

from . import homogeneousmatrix as Hg
from .rigidmotion import RigidMotion


def simplearm():
    """A world holding the 3R planar arm (used by doctests and tests)."""
    from .robots.simplearm import add_simplearm
    w = World()
    add_simplearm(w)
    return w


class NamedObject(object):
    """Anything that may carry a name."""

    def __init__(self, name=None):
        self.name = name

    def __repr__(self):
        if self.name is None:
            return object.__repr__(self)
        return '<{0}.{1} object named "{2}" at "{3}")>'.format(
            self.__class__.__module__, self.__class__.__name__, self.name,
            hex(id(self)))


class DuplicateNameError(Exception):
    pass


class NamedObjectsList(list):
    """A list whose items can also be fetched by name (first match wins)."""

    def __init__(self, iterable=None):
        list.__init__(self)
        if iterable is not None:
            self.extend(iterable)

    def find(self, name):
        return [o for o in self if isinstance(o, NamedObject) and o.name == name]

    def __getitem__(self, index):
        if isinstance(index, str):
            for o in self:
                if isinstance(o, NamedObject) and o.name == index:
                    return o
            raise KeyError('No object named "{0}".'.format(index))
        return list.__getitem__(self, index)

    def as_dict(self):
        out = {}
        for o in self:
            if isinstance(o, NamedObject) and o.name is not None:
                if o.name in out:
                    raise DuplicateNameError()
                out[o.name] = o
        return out


class Frame(object, metaclass=ABCMeta):
    """Abstract frame: a body or a frame rigidly attached to one."""

    @abstractproperty
    def pose(self):
        pass

    @abstractproperty
    def jacobian(self):
        pass

    @abstractproperty
    def djacobian(self):
        pass

    @abstractproperty
    def twist(self):
        pass

    @abstractproperty
    def body(self):
        pass

    @abstractproperty
    def bpose(self):
        pass


class Joint(RigidMotion, NamedObject):
    """Ideal joint between ``frames[0]`` (parent side) and ``frames[1]``."""

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self._frame0 = None
        self._frame1 = None
        self._dof = None          # slice into the world dof vector, set by World.init()

    @abstractproperty
    def ndof(self):
        pass

    @property
    def dof(self):
        if self._dof is None:
            raise ValueError
        return self._dof

    @property
    def frames(self):
        return (self._frame0, self._frame1)

    @property
    def twist(self):
        return dot(self.jacobian, self.gvel)

    @abstractproperty
    def jacobian(self):
        pass

    @abstractproperty
    def djacobian(self):
        pass

    @abstractmethod
    def integrate(self, gvel, dt):
        pass


class LinearConfigurationSpaceJoint(Joint):
    """Joint whose configuration space is R^ndof (``gpos += dt * gvel``)."""

    def __init__(self, gpos=None, gvel=None, name=None):
        k = self.ndof
        self.gpos = zeros(k) if gpos is None else array(gpos, dtype=float).reshape((k,))
        self.gvel = zeros(k) if gvel is None else array(gvel, dtype=float).reshape((k,))
        Joint.__init__(self, name)

    def integrate(self, gvel, dt):
        self.gvel = gvel
        self.gpos += dt * self.gvel


class JointsList(NamedObjectsList):
    """List of joints that also exposes the union of their dofs."""

    def __init__(self, iterable):
        NamedObjectsList.__init__(self, iterable)
        self._init_dof()

    def _init_dof(self):
        dof = slice(0, 0)
        for j in self:
            if not isinstance(j, Joint):
                continue
            assert j.dof.step in (None, 1)
            if isinstance(dof, slice) and dof.stop == j.dof.start:
                dof = slice(dof.start, j.dof.stop)      # still contiguous
            else:
                if isinstance(dof, slice):
                    dof = list(range(dof.start, dof.stop))
                dof.extend(range(j.dof.start, j.dof.stop))
        self._dof = dof

    @property
    def dof(self):
        return self._dof


class Constraint(NamedObject, metaclass=ABCMeta):

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self._is_enabled = True

    def is_enabled(self):
        return self._is_enabled

    def enable(self):
        self._is_enabled = True

    def disable(self):
        self._is_enabled = False

    @abstractmethod
    def init(self, world):
        pass

    @property
    def gforce(self):
        return dot(self.jacobian.T, self._force)

    @abstractproperty
    def jacobian(self):
        pass

    @abstractproperty
    def ndol(self):
        """Number of degrees of "liaison" (6 - ndof of the constrained motion)."""

    @abstractmethod
    def update(self, dt):
        pass

    @abstractmethod
    def is_active(self):
        pass

    @abstractmethod
    def solve(self, vel, admittance, dt):
        pass


class Shape(NamedObject):
    """Geometric primitive attached to a frame, for collision detection."""

    def __init__(self, frame, name=None):
        assert isinstance(frame, Frame)
        self.frame = frame
        NamedObject.__init__(self, name)


class Controller(NamedObject, metaclass=ABCMeta):

    def __init__(self, name=None):
        NamedObject.__init__(self, name)

    @abstractmethod
    def init(self, world):
        pass

    @abstractmethod
    def update(self, dt):
        pass


class Observer(object, metaclass=ABCMeta):

    @abstractmethod
    def init(self, world, timeline):
        pass

    @abstractmethod
    def update(self, dt):
        pass

    @abstractmethod
    def finish(self):
        pass


class World(NamedObject):
    """Tree of bodies and joints rooted at ``ground`` + registered plugins."""

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self.ground = Body('ground')
        self._current_time = 0.
        self._up = array((0., 1., 0.))
        self._controllers = []
        self._constraints = []
        self._subframes = []
        self._shapes = []
        self._ndof = 0
        self._gvel = array([])
        self._mass = array([])
        self._gforce = array([])
        self._viscosity = array([])
        self._nleffects = array([])
        self._impedance = array([])
        self._admittance = array([])
        self._engine = None            # device evaluator, created on first use
        self._constraints_done = False

    # -- iteration ----------------------------------------------------------
    def iterbodies(self):
        """All bodies, ground first, depth-first."""
        yield self.ground
        for b in self.ground.iter_descendant_bodies():
            yield b

    def getbodies(self):
        return NamedObjectsList(self.iterbodies())

    def iterconstraints(self):
        return iter(self._constraints)

    def itersubframes(self):
        return iter(self._subframes)

    def itermovingsubframes(self):
        return (f for f in self._subframes if isinstance(f, MovingSubFrame))

    def iterframes(self):
        for b in self.iterbodies():
            yield b
        for f in self._subframes:
            yield f

    def getframes(self):
        frames = self.getbodies()
        frames.extend(self._subframes)
        return frames

    def itershapes(self):
        return iter(self._shapes)

    def getshapes(self):
        return NamedObjectsList(self._shapes)

    def iterjoints(self):
        """All joints, depth-first (this order defines the dof numbering)."""
        return self.ground.iter_descendant_joints()

    def getjoints(self):
        return JointsList(self.iterjoints())

    # -- construction -------------------------------------------------------
    def add_link(self, frame0, joint, frame1, *args):
        """Attach ``frame1``'s body to the tree through ``joint`` at ``frame0``.

        Several (frame0, joint, frame1) triples may be given at once.
        """
        assert isinstance(frame0, Frame)
        assert isinstance(frame1, Frame)
        assert isinstance(joint, Joint)
        assert joint._frame0 is None
        assert joint._frame1 is None
        assert len(args) % 3 == 0
        if frame1.body.parentjoint is not None:
            raise ValueError(
                "frame1's body already has a parent joint, which means you're "
                "probably trying to create a kinematic loop. Try using a "
                "constraint instead.")
        joint._frame0 = frame0
        joint._frame1 = frame1
        frame1.body.parentjoint = joint
        frame0.body.childrenjoints.append(joint)
        self.register(frame0)
        self.register(frame1)
        if args:
            self.add_link(*args)

    def replace_joint(self, old_joint, *args):
        """``replace_joint(old, new)`` or ``replace_joint(old, f0, j, ..., f1)``."""
        assert isinstance(old_joint, Joint)
        assert old_joint in old_joint._frame0.body.childrenjoints
        assert old_joint is old_joint._frame1.body.parentjoint
        if len(args) == 1:
            self.replace_joint(old_joint, old_joint._frame0, args[0], old_joint._frame1)
            return
        if len(args) == 0 or len(args) % 3 != 0:
            raise RuntimeError()
        body0 = args[0].body
        body1 = args[-1].body
        assert old_joint._frame0.body is body0
        assert old_joint._frame1.body is body1
        body1.parentjoint = None
        old_joint._frame0 = None
        old_joint._frame1 = None
        self.add_link(*args)
        # the new first joint was appended; move it to the old joint's slot
        i = body0.childrenjoints.index(old_joint)
        body0.childrenjoints[i] = body0.childrenjoints.pop()
        self.init()

    def register(self, obj):
        """Register a subframe, shape, constraint or controller."""
        if isinstance(obj, Body):
            pass
        elif isinstance(obj, Joint):
            raise ValueError('Joints should not be registered. Use add_link() instead.')
        elif isinstance(obj, (SubFrame, MovingSubFrame)):
            if obj not in self._subframes:
                self._subframes.append(obj)
        elif isinstance(obj, Shape):
            if obj not in self._shapes:
                self._shapes.append(obj)
            self.register(obj.frame)
        elif isinstance(obj, Constraint):
            if obj not in self._constraints:
                self._constraints.append(obj)
                from .constraints import PointContact
                if isinstance(obj, PointContact):
                    self.register(obj._frames[0])
                    self.register(obj._frames[1])
        elif isinstance(obj, Controller):
            if obj not in self._controllers:
                self._controllers.append(obj)
        else:
            raise ValueError(
                'I do not know how to register objects of type {0}'.format(type(obj)))

    def parse(self, target):
        """Walk the world depth-first calling ``target.init_parse``,
        ``target.register`` and ``target.add_link`` (exporter hook)."""
        seen = set()

        def visit_frame(frame):
            if frame in seen:
                return
            seen.add(frame)
            target.register(frame)
            if isinstance(frame, Body):
                for f in self._subframes:
                    if f not in seen and f.body is frame:
                        visit_frame(f)
            for s in self._shapes:
                if s.frame is frame:
                    target.register(s)

        def visit_joints(joints):
            for j in joints:
                (f0, f1) = j.frames
                target.add_link(f0, j, f1)
                visit_frame(f1.body)
                visit_joints(f1.body.childrenjoints)

        target.init_parse(self.ground, self.up, self.current_time)
        visit_frame(self.ground)
        visit_joints(self.ground.childrenjoints)
        for c in self._constraints:
            target.register(c)
        for a in self._controllers:
            target.register(a)

    def init(self):
        """Number the dofs (DFS joint order) and size the world matrices."""
        n = 0
        for j in self.iterjoints():
            j._dof = slice(n, n + j.ndof)
            n += j.ndof
        self._ndof = n
        self._mass = zeros((n, n))
        self._nleffects = zeros((n, n))
        self._viscosity = zeros((n, n))
        self._controller_viscosity = zeros((n, n))
        self._gforce = zeros(n)
        self._gvel = zeros(n)
        for j in self.iterjoints():
            self._gvel[j.dof] = j.gvel[:]
            j.gvel = self._gvel[j.dof]           # joints alias the world vector
        for c in self._constraints:
            c.init(self)
        for a in self._controllers:
            a.init(self)
        self._constraints_done = False

    # -- read-only state ----------------------------------------------------
    @property
    def current_time(self):
        return self._current_time

    @property
    def up(self):
        return self._up

    @property
    def mass(self):
        return self._mass

    @property
    def viscosity(self):
        return self._viscosity

    @property
    def nleffects(self):
        return self._nleffects

    @property
    def ndof(self):
        return self._ndof

    @property
    def gvel(self):
        return self._gvel.copy()

    @property
    def gforce(self):
        return self._gforce.copy()

    # -- the step: all four stages run on the device -------------------------
    def _device(self):
        if self._engine is None:
            from ._engine import SingleWorldEngine
            self._engine = SingleWorldEngine(self)
        return self._engine

    def update_geometric(self):
        """Forward geometric model: refresh every ``Body.pose``."""
        self._device().run(self, 'geometric', None)

    def update_dynamic(self):
        """Forward kinematic + dynamic model.

        Refreshes each body's ``pose, jacobian, djacobian, twist, nleffects``
        and the world ``mass, viscosity, nleffects`` matrices
        (M = sum J^T M_b J, B = sum J^T B_b J, N = sum J^T (M_b dJ + N_b J)).
        """
        self._device().run(self, 'dynamic', None)
        self._constraints_done = False

    def update_controllers(self, dt):
        """Z = M/dt + B + N - sum Z_a ; Y = Z^-1 ; gforce = sum controllers."""
        assert dt > 0
        self._device().run(self, 'controllers', dt)
        self._constraints_done = False

    def update_constraints(self, dt):
        """20-sweep Gauss-Seidel over the active constraints; adds their
        generalized forces to ``gforce``."""
        assert dt > 0
        self._device().run(self, 'constraints', dt)
        self._constraints_done = True

    def integrate(self, dt):
        """gvel <- Y (M gvel/dt + gforce); every joint integrates its gpos."""
        assert dt > 0
        self._device().run(self, 'integrate', dt)
        self._current_time += dt
        self._constraints_done = False


class _SubFrame(NamedObject, Frame):
    """Frame rigidly attached to a body at the constant offset ``bpose``."""

    def __init__(self, body, bpose=None, name=None):
        if bpose is None:
            bpose = eye(4)
        NamedObject.__init__(self, name)
        assert Hg.ishomogeneousmatrix(bpose)
        self._bpose = bpose
        if not isinstance(body, Body):
            raise ValueError(
                "The ``body`` argument must be an instance of the ``Boby`` class")
        self._body = body

    @property
    def pose(self):
        return dot(self._body.pose, self._bpose)

    @property
    def twist(self):
        return dot(Hg.iadjoint(self._bpose), self._body._twist)

    @property
    def jacobian(self):
        return dot(Hg.iadjoint(self._bpose), self._body._jacobian)

    @property
    def djacobian(self):
        return dot(Hg.iadjoint(self._bpose), self._body._djacobian)

    @property
    def body(self):
        return self._body


class SubFrame(_SubFrame):
    @property
    def bpose(self):
        return self._bpose.copy()


class MovingSubFrame(_SubFrame):
    @property
    def bpose(self):
        return self._bpose.copy()

    @bpose.setter
    def bpose(self, bpose):
        assert Hg.ishomogeneousmatrix(bpose)
        self._bpose[:] = bpose


class Body(NamedObject, Frame):
    """Rigid body: 6x6 ``mass`` and ``viscosity`` about its own frame."""

    def __init__(self, name=None, mass=None, viscosity=None):
        NamedObject.__init__(self, name)
        self.parentjoint = None
        self.childrenjoints = []
        self.mass = zeros((6, 6)) if mass is None else mass
        self.viscosity = zeros((6, 6)) if viscosity is None else viscosity
        self._pose = None
        self._jacobian = None
        self._djacobian = None
        self._twist = None
        self._nleffects = None

    def iter_descendant_bodies(self):
        for j in self.childrenjoints:
            child = j._frame1.body
            yield child
            for b in child.iter_descendant_bodies():
                yield b

    def iter_ancestor_bodies(self):
        j = self.parentjoint
        while j is not None:
            parent = j._frame0.body
            yield parent
            j = parent.parentjoint

    def iter_descendant_joints(self):
        for j in self.childrenjoints:
            yield j
            for jj in j._frame1.body.iter_descendant_joints():
                yield jj

    def iter_ancestor_joints(self):
        j = self.parentjoint
        while j is not None:
            yield j
            j = j._frame0.body.parentjoint

    @property
    def pose(self):
        return self._pose

    @property
    def jacobian(self):
        return self._jacobian

    @property
    def djacobian(self):
        return self._djacobian

    @property
    def twist(self):
        return self._twist

    @property
    def nleffects(self):
        return self._nleffects

    @property
    def bpose(self):
        return eye(4)

    @property
    def body(self):
        return self


def simulate(world, timeline, observers=()):
    """Run ``len(timeline) - 1`` steps; ``dt`` is taken from the timeline.

    Observers are updated between ``update_constraints`` and ``integrate``:
    they see the state at time t and the forces for [t, t+dt].
    """
    world._current_time = timeline[0]
    world.init()
    for obs in observers:
        obs.init(world, timeline)
    for next_time in timeline[1:]:
        dt = next_time - world._current_time
        world.update_dynamic()
        world.update_controllers(dt)
        world.update_constraints(dt)
        for obs in observers:
            obs.update(dt)
        world.integrate(dt)
    for obs in observers:
        obs.finish()
