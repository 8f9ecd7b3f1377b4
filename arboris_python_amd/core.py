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


def _abstract(*names):
    """Read-only abstract properties ``names`` for a class body: ``locals().update(_abstract("pose", ...))``."""
    def make(name):
        def getter(self):
            raise NotImplementedError(name)
        getter.__name__ = name
        return abstractproperty(getter)
    return dict((n, make(n)) for n in names)


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
    """Abstract frame: a body or a frame rigidly attached to one (pose, jacobian, djacobian, twist, body, bpose)."""
    locals().update(_abstract("pose", "jacobian", "djacobian", "twist", "body", "bpose"))


class Joint(RigidMotion, NamedObject):
    """Ideal joint between ``frames[0]`` (parent side) and ``frames[1]``; a concrete joint provides ``ndof``,
    ``jacobian``, ``djacobian`` and ``integrate(gvel, dt)``."""
    locals().update(_abstract("ndof", "jacobian", "djacobian"))

    @abstractmethod
    def integrate(self, gvel, dt):
        raise NotImplementedError("integrate")

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self._frame0 = None
        self._frame1 = None
        self._dof = None          # slice into the world dof vector, set by World.init()

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
    """A kinematic constraint.  Concrete classes provide ``jacobian``, ``ndol`` (degrees of "liaison": 6 minus the
    dofs of the constrained motion), ``init(world)``, ``update(dt)``, ``is_active()`` and
    ``solve(vel, admittance, dt)``; the generalized force is ``jacobian.T @ force``."""
    locals().update(_abstract("jacobian", "ndol"))

    @abstractmethod
    def init(self, world):
        raise NotImplementedError("init")

    @abstractmethod
    def update(self, dt):
        raise NotImplementedError("update")

    @abstractmethod
    def is_active(self):
        raise NotImplementedError("is_active")

    @abstractmethod
    def solve(self, vel, admittance, dt):
        raise NotImplementedError("solve")

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self._is_enabled = True

    def enable(self, on=True):
        self._is_enabled = bool(on)

    def disable(self):
        self.enable(False)

    def is_enabled(self):
        return self._is_enabled

    @property
    def gforce(self):
        return dot(self.jacobian.T, self._force)


class Shape(NamedObject):
    """Geometric primitive attached to a frame, for collision detection."""

    def __init__(self, frame, name=None):
        assert isinstance(frame, Frame)
        self.frame = frame
        NamedObject.__init__(self, name)


class Controller(NamedObject, metaclass=ABCMeta):
    """A controller: ``init(world)`` once, then ``update(dt)`` returns (gforce, impedance) every step."""

    def __init__(self, name=None):
        NamedObject.__init__(self, name)

    @abstractmethod
    def init(self, world):
        raise NotImplementedError("init")

    @abstractmethod
    def update(self, dt):
        raise NotImplementedError("update")


class Observer(object, metaclass=ABCMeta):
    """Watches a simulation: ``init(world, timeline)``, ``update(dt)`` before every step, ``finish()``."""

    @abstractmethod
    def init(self, world, timeline):
        raise NotImplementedError("init")

    @abstractmethod
    def update(self, dt):
        raise NotImplementedError("update")

    @abstractmethod
    def finish(self):
        raise NotImplementedError("finish")


class _Registry(object):
    """Insertion-ordered, duplicate-free collections of the things a world knows about besides its tree:
    sub-frames, shapes, constraints, controllers.  ``kind_of`` maps an object to its collection."""

    KINDS = ('subframes', 'shapes', 'constraints', 'controllers')

    def __init__(self):
        for kind in self.KINDS:
            setattr(self, kind, [])

    def add(self, kind, obj):
        """Append ``obj`` to its collection unless it is there already; tells whether it was added."""
        items = getattr(self, kind)
        if any(o is obj for o in items):
            return False
        items.append(obj)
        return True


class World(NamedObject):
    """Tree of bodies and joints rooted at ``ground`` + registered plugins."""

    def __init__(self, name=None):
        NamedObject.__init__(self, name)
        self.ground = Body('ground')
        self._current_time = 0.
        self._up = array((0., 1., 0.))
        self._reg = _Registry()
        self._ndof = 0
        empty = array([])
        self._gvel = self._gforce = empty
        self._mass = self._viscosity = self._nleffects = empty
        self._impedance = self._admittance = empty
        self._engine = None            # device evaluator, created on first use
        self._constraints_done = False

    # the collections under their historical attribute names (observers, exporters and tests read them)
    _subframes = property(lambda self: self._reg.subframes)
    _shapes = property(lambda self: self._reg.shapes)
    _constraints = property(lambda self: self._reg.constraints)
    _controllers = property(lambda self: self._reg.controllers)

    # -- iteration ----------------------------------------------------------
    def iterbodies(self):
        """All bodies, ground first, depth-first."""
        yield self.ground
        yield from self.ground.iter_descendant_bodies()

    def getbodies(self):
        return NamedObjectsList(self.iterbodies())

    def iterconstraints(self):
        return iter(self._reg.constraints)

    def itersubframes(self):
        return iter(self._reg.subframes)

    def itermovingsubframes(self):
        return (f for f in self._reg.subframes if isinstance(f, MovingSubFrame))

    def iterframes(self):
        yield from self.iterbodies()
        yield from self._reg.subframes

    def getframes(self):
        return NamedObjectsList(self.iterframes())

    def itershapes(self):
        return iter(self._reg.shapes)

    def getshapes(self):
        return NamedObjectsList(self._reg.shapes)

    def iterjoints(self):
        """All joints, depth-first (this order defines the dof numbering)."""
        return self.ground.iter_descendant_joints()

    def getjoints(self):
        return JointsList(self.iterjoints())

    # -- construction -------------------------------------------------------
    def _attach(self, frame0, joint, frame1):
        """One edge of the tree: ``frame1``'s body hangs from ``frame0``'s body through ``joint``."""
        for f in (frame0, frame1):
            assert isinstance(f, Frame)
        assert isinstance(joint, Joint)
        assert joint.frames == (None, None)
        child, parent = frame1.body, frame0.body
        if child.parentjoint is not None:
            raise ValueError(
                "frame1's body already has a parent joint, which means you're "
                "probably trying to create a kinematic loop. Try using a "
                "constraint instead.")
        joint._frame0, joint._frame1 = frame0, frame1
        child.parentjoint = joint
        parent.childrenjoints.append(joint)
        self.register(frame0)
        self.register(frame1)

    def add_link(self, frame0, joint, frame1, *args):
        """Attach ``frame1``'s body to the tree through ``joint`` at ``frame0``.

        Several (frame0, joint, frame1) triples may be given at once.
        """
        assert len(args) % 3 == 0
        chain = (frame0, joint, frame1) + args
        for k in range(0, len(chain), 3):
            self._attach(*chain[k:k + 3])

    def replace_joint(self, old_joint, *args):
        """``replace_joint(old, new)`` or ``replace_joint(old, f0, j, ..., f1)``."""
        assert isinstance(old_joint, Joint)
        parent, child = old_joint._frame0.body, old_joint._frame1.body
        assert old_joint is child.parentjoint
        slot = [k for k, j in enumerate(parent.childrenjoints) if j is old_joint]
        assert len(slot) == 1
        if len(args) == 1:                       # same frames, another joint
            args = (old_joint._frame0, args[0], old_joint._frame1)
        if len(args) == 0 or len(args) % 3 != 0:
            raise RuntimeError()
        assert args[0].body is parent
        assert args[-1].body is child
        # cut the old edge, hang the new chain, and give its first joint the old joint's place among the
        # parent's children (the dof numbering follows that order)
        child.parentjoint = None
        old_joint._frame0 = old_joint._frame1 = None
        self.add_link(*args)
        first_new = parent.childrenjoints.pop()
        parent.childrenjoints[slot[0]] = first_new
        self.init()

    def register(self, obj):
        """Register a subframe, shape, constraint or controller."""
        if isinstance(obj, Joint):
            raise ValueError('Joints should not be registered. Use add_link() instead.')
        if isinstance(obj, Body):
            return                               # bodies are reached through the tree
        if isinstance(obj, (SubFrame, MovingSubFrame)):
            self._reg.add('subframes', obj)
        elif isinstance(obj, Shape):
            self._reg.add('shapes', obj)
            self.register(obj.frame)
        elif isinstance(obj, Constraint):
            if self._reg.add('constraints', obj):
                from .constraints import PointContact
                if isinstance(obj, PointContact):    # its two contact frames move with the contact point
                    for f in obj._frames:
                        self.register(f)
        elif isinstance(obj, Controller):
            self._reg.add('controllers', obj)
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
                for f in self._reg.subframes:
                    if f not in seen and f.body is frame:
                        visit_frame(f)
            for s in self._reg.shapes:
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
        for plugin in self._reg.constraints + self._reg.controllers:
            target.register(plugin)

    def init(self):
        """Number the dofs (DFS joint order) and size the world matrices."""
        joints = list(self.iterjoints())
        start = 0
        for j in joints:
            j._dof = slice(start, start + j.ndof)
            start = j._dof.stop
        n = self._ndof = start
        for name in ('_mass', '_nleffects', '_viscosity', '_controller_viscosity'):
            setattr(self, name, zeros((n, n)))
        self._gforce = zeros(n)
        # the world owns the generalized velocity; every joint's ``gvel`` becomes a view of its slice
        self._gvel = zeros(n)
        for j in joints:
            view = self._gvel[j._dof]
            view[:] = j.gvel
            j.gvel = view
        for plugin in self._reg.constraints + self._reg.controllers:
            plugin.init(self)
        self._constraints_done = False

    # -- read-only state ----------------------------------------------------
    def _view(attribute, copy=False):        # noqa: N805 -- builds the read-only properties below
        if copy:
            return property(lambda self: getattr(self, attribute).copy())
        return property(lambda self: getattr(self, attribute))

    current_time = _view("_current_time")
    up = _view("_up")
    ndof = _view("_ndof")
    mass, viscosity, nleffects = _view("_mass"), _view("_viscosity"), _view("_nleffects")
    gvel, gforce = _view("_gvel", copy=True), _view("_gforce", copy=True)
    del _view

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
    """Frame rigidly attached to a body at the constant offset ``bpose``: everything the frame reports is the
    body's quantity carried through ``Ad(bpose^-1)`` (pose: multiplied by ``bpose``)."""

    def __init__(self, body, bpose=None, name=None):
        NamedObject.__init__(self, name)
        if not isinstance(body, Body):
            raise ValueError("The ``body`` argument must be an instance of the ``Body`` class")
        bpose = eye(4) if bpose is None else bpose
        assert Hg.ishomogeneousmatrix(bpose)
        self._body, self._bpose = body, bpose

    def _carried(self, body_quantity):
        return dot(Hg.iadjoint(self._bpose), body_quantity)

    pose = property(lambda self: dot(self._body.pose, self._bpose))
    twist = property(lambda self: self._carried(self._body._twist))
    jacobian = property(lambda self: self._carried(self._body._jacobian))
    djacobian = property(lambda self: self._carried(self._body._djacobian))
    body = property(lambda self: self._body)


class SubFrame(_SubFrame):
    bpose = property(lambda self: self._bpose.copy())


class MovingSubFrame(_SubFrame):
    """A sub-frame whose offset may be moved (contact frames follow the contact point)."""

    def _get_bpose(self):
        return self._bpose.copy()

    def _set_bpose(self, bpose):
        assert Hg.ishomogeneousmatrix(bpose)
        self._bpose[:] = bpose

    bpose = property(_get_bpose, _set_bpose)


class Body(NamedObject, Frame):
    """Rigid body: 6x6 ``mass`` and ``viscosity`` about its own frame."""

    _STATE = ('pose', 'jacobian', 'djacobian', 'twist', 'nleffects')      # filled by World.update_*

    def __init__(self, name=None, mass=None, viscosity=None):
        NamedObject.__init__(self, name)
        self.parentjoint = None
        self.childrenjoints = []
        self.mass = zeros((6, 6)) if mass is None else mass
        self.viscosity = zeros((6, 6)) if viscosity is None else viscosity
        for field in self._STATE:
            setattr(self, '_' + field, None)

    # depth-first walks of the tree below / above this body (children in insertion order: the dof order)
    def iter_descendant_joints(self):
        pending = list(reversed(self.childrenjoints))
        while pending:
            j = pending.pop()
            yield j
            pending.extend(reversed(j._frame1.body.childrenjoints))

    def iter_descendant_bodies(self):
        return (j._frame1.body for j in self.iter_descendant_joints())

    def iter_ancestor_joints(self):
        j = self.parentjoint
        while j is not None:
            yield j
            j = j._frame0.body.parentjoint

    def iter_ancestor_bodies(self):
        return (j._frame0.body for j in self.iter_ancestor_joints())

    pose = property(lambda self: self._pose)
    jacobian = property(lambda self: self._jacobian)
    djacobian = property(lambda self: self._djacobian)
    twist = property(lambda self: self._twist)
    nleffects = property(lambda self: self._nleffects)
    bpose = property(lambda self: eye(4))
    body = property(lambda self: self)


def simulate(world, timeline, observers=()):
    """Run ``len(timeline) - 1`` steps; ``dt`` is taken from the timeline.

    Observers are updated between ``update_constraints`` and ``integrate``:
    they see the state at time t and the forces for [t, t+dt].
    """
    def tell(event, *args):
        for obs in observers:
            getattr(obs, event)(*args)

    world._current_time = timeline[0]
    world.init()
    tell('init', world, timeline)
    for next_time in timeline[1:]:
        dt = next_time - world._current_time
        world.update_dynamic()
        world.update_controllers(dt)
        world.update_constraints(dt)
        tell('update', dt)
        world.integrate(dt)
    tell('finish')
