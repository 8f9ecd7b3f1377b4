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


def human36_world(contacts=0, gravity=True, friction_coeff=0.6, pd=False):
    """human36 (+ ground plane and 4, 6 or 8 SoftFingerContacts; 6 = three per foot, the reference's points 0 1 2 4 5 7).
    ``pd``: a ProportionalDerivativeController on every hinge dof that holds the standing pose (kp 20 N m / rad, kd 2): a
    human36 OUTSIDE the model class of the specialised kernels (bench.py's ``model_classes`` legs)."""
    assert contacts in (0, 4, 6, 8)
    w = World()
    if contacts:
        add_groundplane(w)
    add_human36(w)
    if gravity:
        w.register(WeightController())
    if pd:
        from .controllers import ProportionalDerivativeController
        from .joints import LinearConfigurationSpaceJoint
        js = [j for j in w.getjoints() if isinstance(j, LinearConfigurationSpaceJoint)]
        n = sum(j.ndof for j in js)
        w.register(ProportionalDerivativeController(js, kp=20. * np.eye(n), kd=2. * np.eye(n)))
    if contacts:
        for i, c in enumerate(get_all_contacts(w, friction_coeff=friction_coeff)):
            if contacts == 8 or (contacts == 6 and i not in (3, 6)) or (contacts == 4 and c._shapes[1].name in FOUR_CONTACTS):
                w.register(c)
    w.init()
    return w


def human36_and_objects_world(nobjects=4, friction_coeff=0.6):
    """human36 on the floor (its four heel / toe-tip contacts) beside ``nobjects`` free boxes, each carrying a ball that
    touches the floor: 42 + 6 nobjects dofs -- with four objects 66, PAST the 64 lanes of one wavefront: the smallest
    everyday scene that needs the wide kernels (csrc/arb_wide_kernel.h; the reference allocates any ndof, core.py:608-635)."""
    from .core import Body
    from .joints import FreeJoint
    from .shapes import Sphere
    from . import massmatrix, homogeneousmatrix as Hg
    w = World()
    add_groundplane(w)
    add_human36(w)
    for k in range(nobjects):
        he = (0.10 + 0.02 * k, 0.08, 0.12)
        body = Body(name="Box%d" % k, mass=massmatrix.box(he, 2.0 + k))
        j = FreeJoint(name="BoxRoot%d" % k)
        j.gpos = Hg.transl(0.6 + 0.5 * k, 0.13 + 0.01 * k, 0.4 - 0.3 * k)
        w.add_link(w.ground, j, body)
        w.register(Sphere(body, 0.12, name="Box%d ball" % k))
    w.register(WeightController())
    for c in get_all_contacts(w, friction_coeff=friction_coeff):
        s0, s1 = c._shapes
        if type(s0).__name__ == "Plane" and (s1.name in FOUR_CONTACTS or str(s1.name).endswith(" ball")):
            w.register(c)
    w.init()
    return w


def human36_and_balls_world(nballs=3, friction_coeff=0.6):
    """human36 beside ``nballs`` free balls on the ground plane with EVERY pair of shapes ``get_all_contacts`` finds
    (constraints.py:840-875) registered: plane / point (the eight foot points), plane / ball, ball / point, ball / ball --
    8 + n + 8 n + n (n - 1) / 2 SoftFingerContacts (three balls: 38 on 60 dofs; eight: 108 on 90 dofs), of which a step has
    ten to thirty active: the reference's idiom for a scene, and what the wide kernels' slots are for."""
    from .core import Body
    from .joints import FreeJoint
    from .shapes import Sphere
    from . import massmatrix, homogeneousmatrix as Hg
    w = World()
    add_groundplane(w)
    add_human36(w)
    for k in range(nballs):
        body = Body(name="Ball%d" % k, mass=massmatrix.sphere(0.1, 1.0 + k))
        j = FreeJoint(name="BallRoot%d" % k)
        # (rows of balls in front of the feet, touching one another)
        j.gpos = Hg.transl(0.12 + 0.19 * (k % 4), 0.105, 0.05 * (k % 4) + 0.21 * (k // 4))
        w.add_link(w.ground, j, body)
        w.register(Sphere(body, 0.1, name="Ball%d" % k))
    w.register(WeightController())
    for c in get_all_contacts(w, friction_coeff=friction_coeff):
        w.register(c)
    w.init()
    return w


def flat(world):
    """FlatModel of an initialised world."""
    return flatten_world(world)[0]


class _NS(object):
    pass


def _own_namespace():
    from . import core, shapes, joints, massmatrix, homogeneousmatrix
    from .robots.simpleshapes import add_sphere
    W = _NS()
    W.World, W.Body, W.SubFrame, W.Hg = core.World, core.Body, core.SubFrame, homogeneousmatrix
    W.Box, W.Sphere, W.Point = shapes.Box, shapes.Sphere, shapes.Point
    W.add_sphere, W.add_groundplane = add_sphere, add_groundplane
    W.WeightController, W.get_all_contacts, W.FreeJoint = WeightController, get_all_contacts, joints.FreeJoint
    W.massmatrix = massmatrix
    return W


def shape_scenes(W=None):
    """Four small scenes, one per remaining narrow-phase pair (collisions.py:27-64), built from
    the class namespace `W`: this package by default, the reference's classes when
    tools/gen_golden.py records tests/golden/g7_shapes.npz."""
    if W is None:
        W = _own_namespace()
    out = {}
    # (1) ball rolling on the ground plane: Plane / Sphere with a non-zero radius
    w = W.World()
    W.add_groundplane(w)
    W.add_sphere(w, radius=0.3, mass=2., name="ball")
    j = w.ground.childrenjoints[0]
    j.gpos = W.Hg.transl(0., 0.31, 0.)
    j.gvel = np.array([0., 0., -2., 1., 0., 0.2])
    w.register(W.WeightController())
    for c in W.get_all_contacts(w, friction_coeff=0.5):
        w.register(c)
    w.init()
    out["plane_ball"] = w
    # (2) ball dropped on the edge region of a box fixed to the ground: Box / Sphere
    w = W.World()
    f = W.SubFrame(w.ground, W.Hg.transl(0., 0.2, 0.), "box frame")
    w.register(W.Box(f, (0.5, 0.2, 0.5), "box"))
    W.add_sphere(w, radius=0.1, mass=1., name="ball")
    j = w.ground.childrenjoints[0]
    j.gpos = W.Hg.transl(0.47, 0.51, 0.1)
    j.gvel = np.array([0.3, 0., 0., 0.4, -0.5, 0.])
    w.register(W.WeightController())
    for c in W.get_all_contacts(w, friction_coeff=0.4):
        w.register(c)
    w.init()
    out["box_ball"] = w
    # (3) two free balls colliding, no gravity: Sphere / Sphere, both bodies moving
    w = W.World()
    W.add_sphere(w, radius=0.2, mass=1., name="a")
    W.add_sphere(w, radius=0.3, mass=3., name="b")
    ja, jb = w.ground.childrenjoints
    ja.gpos = W.Hg.transl(0., 0., 0.)
    ja.gvel = np.array([0., 0.5, 0., 2., 0., 0.])
    jb.gpos = W.Hg.transl(0.56, 0.12, 0.03)
    jb.gvel = np.array([0., 0., 0., 0., 0., 0.])
    for c in W.get_all_contacts(w, friction_coeff=0.3):
        w.register(c)
    w.init()
    out["ball_ball"] = w
    # (4) a body carrying a Point dropped on a big sphere fixed to the ground: Sphere / Point
    w = W.World()
    w.register(W.Sphere(w.ground, 0.5, "dome"))
    body = W.Body(name="probe", mass=W.massmatrix.box((0.1, 0.1, 0.1), 1.5))
    w.add_link(w.ground, W.FreeJoint(), body)
    tip = W.SubFrame(body, W.Hg.transl(0., -0.1, 0.), "tip")
    w.register(W.Point(tip, "tip point"))
    j = w.ground.childrenjoints[0]
    j.gpos = W.Hg.transl(0.1, 0.615, -0.05)
    j.gvel = np.array([0., 0., 0., 0., -0.3, 0.])
    w.register(W.WeightController())
    for c in W.get_all_contacts(w, friction_coeff=0.6):
        w.register(c)
    w.init()
    out["dome_point"] = w
    return out
