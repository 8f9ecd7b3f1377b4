"""Anthropometric 42-dof humanoid ("human36" of the HuMAnS toolbox).

Same model as the reference's arboris/robots/human36.py (lengths :54-153,
height check :155-183, bodies/joints/tags/foot points :187-399), rebuilt here
from data tables: segment lengths as fractions of the body height, 17 bodies
(mass fraction, centre of mass, radii of gyration), 17 joints in depth-first
order (FreeJoint pelvis + hinge compounds, 42 dofs), 28 anatomical landmarks
("tags") and one ``Point`` shape on each of the 8 foot landmarks.
"""
import numpy as np

from ..core import World, Body, SubFrame, NamedObjectsList
from .. import homogeneousmatrix as Hg
from ..joints import (FreeJoint, RzRyRxJoint, RzRyJoint, RzRxJoint, RyRxJoint,
                      RzJoint)
from ..shapes import Point

# length name -> fraction of the height (HuMAnS ``SetModelSize``)
_SYMMETRIC = {          # defined for both sides, suffixed R / L
    "yfoot": 0.0222, "ytibia": 0.2493, "yfemur": 0.2425,
    "ysternoclav": 0.0980, "xsternoclav": 0.1052,
    "yshoulder": 0.0104, "xshoulder": 0.0526,
    "yhumerus": 0.1618, "yforearm": 0.1544, "yhand": 0.1091,
    "xfoot": 0.1482, "xheel": 0.0248,
}
_CENTRAL = {"yvT10": 0.2075, "xvT10": 0.0526, "zhip": 0.1002,
            "yvC7": 0.139, "yhead": 0.1395}
_HALF_INCH = 0.5 * 0.0254


def anat_lengths_from_height(height):
    """Anatomical lengths (dict) scaled from the total ``height`` in metres."""
    L = {}
    for key, frac in _CENTRAL.items():
        L[key] = float(frac * height)
    for key, frac in _SYMMETRIC.items():
        for side in "RL":
            L[key + side] = float(frac * height)
    for side in "RL":
        L["zsternoclav" + side] = float(_HALF_INCH)
        L["zshoulder" + side] = float(0.1295 * height - _HALF_INCH)
    return L


def height_from_anat_lengths(lengths):
    """Total height; raises ValueError when the legs differ in length."""
    legs = [lengths['yfoot' + s] + lengths['ytibia' + s] + lengths['yfemur' + s]
            for s in "RL"]
    if legs[1] != legs[0]:
        raise ValueError("The legs have different lengths")
    return legs[1] + lengths['yvT10'] + lengths['yvC7'] + lengths['yhead']


def _body_table(L):
    """(name, mass fraction, centre of mass, radii of gyration) per body."""
    a = np.array
    rows = [("LPT", 0.275, [0, 0.5108 * L['yvT10'], 0],
             a([0.2722, 0.2628, 0.226]) * L['yvT10'])]
    for s in "RL":
        rows += [
            ("Thigh" + s, 0.1416, [0, -0.4095 * L['yfemur' + s], 0],
             a([0.329, 0.149, 0.329]) * L['yfemur' + s]),
            ("Shank" + s, 0.0433, [0, -0.4459 * L['ytibia' + s], 0],
             a([0.255, 0.103, 0.249]) * L['ytibia' + s]),
            ("Foot" + s, 0.0137,
             [0.4415 * L['xfoot' + s] - L['xheel' + s], -L['yfoot' + s] / 2., 0.],
             a([0.124, 0.257, 0.245]) * L['xfoot' + s]),
        ]
    rows.append(("UPT", 0.1596,
                 [(L['xsternoclavR'] + L['xsternoclavL']) / 4.,
                  0.7001 * (L['ysternoclavR'] + L['ysternoclavL']) / 2., 0.],
                 a([0.716, 0.659, 0.454]) * L['ysternoclavR']))
    hand_gyration = {"R": [0.235, 0.184, 0.288], "L": [0.288, 0.184, 0.235]}
    for s in "RL":
        rows += [
            ("Scapula" + s, 0., [0., 0., 0.], a([0., 0., 0.])),
            ("Arm" + s, 0.0271, [0., -0.5772 * L['yhumerus' + s], 0.],
             a([0.285, 0.158, 0.269]) * L['yhumerus' + s]),
            ("Forearm" + s, 0.0162, [0., -0.4574 * L['yforearm' + s], 0.],
             a([0.276, 0.121, 0.265]) * L['yforearm' + s]),
            ("Hand" + s, 0.0061, [0, -0.3691 * L['yhand' + s], 0],
             a(hand_gyration[s]) * L['yhand' + s]),
        ]
    rows.append(("Head", 0.0694, [0, 0.4998 * L['yhead'], 0],
                 a([0.303, 0.261, 0.315]) * L['yhead']))
    return rows


def _link_table(L):
    """(parent body or None for ground, anchor translation, joint class, child)
    in the order that defines the dof numbering."""
    rows = [(None, (0, L['yfootL'] + L['ytibiaL'] + L['yfemurL'], 0), FreeJoint, 'LPT')]
    for s, sign in (("R", 1.), ("L", -1.)):
        rows += [
            ('LPT', (0, 0, sign * L['zhip'] / 2.), RzRyRxJoint, 'Thigh' + s),
            ('Thigh' + s, (0, -L['yfemur' + s], 0), RzJoint, 'Shank' + s),
            ('Shank' + s, (0, -L['ytibia' + s], 0), RzRxJoint, 'Foot' + s),
        ]
    rows.append(('LPT', (-L['xvT10'], L['yvT10'], 0), RzRyRxJoint, 'UPT'))
    for s, sign in (("R", 1.), ("L", -1.)):
        rows += [
            ('UPT', (L['xsternoclav' + s], L['ysternoclav' + s],
                     sign * L['zsternoclav' + s]), RyRxJoint, 'Scapula' + s),
            ('Scapula' + s, (-L['xshoulder' + s], L['yshoulder' + s],
                             sign * L['zshoulder' + s]), RzRyRxJoint, 'Arm' + s),
            ('Arm' + s, (0, -L['yhumerus' + s], 0), RzRyJoint, 'Forearm' + s),
            ('Forearm' + s, (0, -L['yforearm' + s], 0), RzRxJoint, 'Hand' + s),
        ]
    rows.append(('UPT', (L['xvT10'], L['yvC7'], 0), RzRyRxJoint, 'Head'))
    return rows


def _tag_table(L, h):
    """(landmark name, body, local position); HuMAnS names are kept verbatim
    (including their inconsistent capitalisation)."""
    toe = 1e-4 * h          # HuMAnS compatibility offset on the toe tips
    return [
        ('Right foot toe tip', 'FootR', [L['xfootR'] - L['xheelR'] + toe, -L['yfootR'], 0.]),
        ('Right foot heel', 'FootR', [-L['xheelR'], -L['yfootR'], 0.]),
        ('Right foot phalange 5', 'FootR', [0.0662 * h, -L['yfootR'], 0.0305 * h]),
        ('Right foot Phalange 1', 'FootR', [0.0662 * h, -L['yfootR'], -0.0305 * h]),
        ('Right foot lateral malleolus', 'ShankR', [0., -L['ytibiaR'], 0.0249 * h]),
        ('Femoral lateral epicondyle', 'ThighR', [0., -L['yfemurR'], 0.0290 * h]),
        ('Right great trochanter', 'ThighR', [0., 0., 0.0941 * h - L['zhip'] / 2.]),
        ('Right iliac crest', 'LPT', [0.0271 * h, 0.0366 * h, 0.0697 * h]),
        ('Left foot toe tip', 'FootL', [L['xfootL'] - L['xheelL'] + toe, -L['yfootL'], 0.]),
        ('Left foot heel', 'FootL', [-L['xheelL'], -L['yfootL'], 0.]),
        ('Left foot phalange 5', 'FootL', [0.0662 * h, -L['yfootL'], -0.0305 * h]),
        ('Left foot phalange 1', 'FootL', [0.0662 * h, -L['yfootL'], 0.0305 * h]),
        ('Left foot lateral malleolus', 'ShankL', [0, -L['ytibiaL'], -0.0249 * h]),
        ('Left femoral lateral epicondyle', 'ThighL', [0, -L['yfemurL'], -0.0290 * h]),
        ('Left great trochanter', 'ThighL', [0, 0, -0.0941 * h + L['zhip'] / 2.]),
        ('Left iliac crest', 'LPT', [0.0271 * h, 0.0366 * h, -0.0697 * h]),
        ('Substernale (Xyphoid)', 'UPT', [0.1219 * h, 0, 0]),
        ('Suprasternale', 'UPT', [(L['xsternoclavL'] + L['xsternoclavL']) / 2.,
                                  (L['ysternoclavL'] + L['ysternoclavL']) / 2., 0]),
        ('Right acromion', 'ScapulaR', [-L['xshoulderR'], 0.0198 * h + L['yshoulderR'],
                                        L['zshoulderR']]),
        ('Right humeral lateral epicondyle (radiale)', 'ArmR',
         [0., -L['yhumerusR'], 0.0211 * h]),
        ('Right stylion', 'ForearmR', [0., -0.1533 * h, 0.0331 * h]),
        ('Right 3rd dactylion', 'HandR', [0., -L['yhandR'], 0.]),
        ('Left acromion', 'ScapulaL', [-L['xshoulderL'], 0.0198 * h + L['yshoulderL'],
                                       -L['zshoulderL']]),
        ('Left humeral lateral epicondyle (radiale)', 'ArmL',
         [0., -L['yhumerusL'], -0.0211 * h]),
        ('Left stylion', 'ForearmL', [0., -0.1533 * h, -0.0331 * h]),
        ('Left 3rd dactylion', 'HandL', [0., -L['yhandL'], 0.]),
        ('Cervicale', 'UPT', [-0.0392 * 0. + L['xvT10'], L['yvC7'], 0.]),
        ('Vertex', 'Head', [0., L["yhead"], 0.]),
    ]


FOOT_POINTS = ('Right foot toe tip', 'Right foot heel',
               'Right foot phalange 5', 'Right foot Phalange 1',
               'Left foot toe tip', 'Left foot heel',
               'Left foot phalange 5', 'Left foot phalange 1')


def add_human36(world, height=1.741, mass=73, anat_lengths=None, name=''):
    """Add the humanoid to ``world`` (names prefixed by ``name``) and call
    ``world.init()``.  ``height`` is ignored when ``anat_lengths`` is given."""
    assert isinstance(world, World)
    L = anat_lengths_from_height(height) if anat_lengths is None else anat_lengths
    h = height_from_anat_lengths(L)
    prefix = name

    bodies = NamedObjectsList()
    for (bname, fraction, com, gyration) in _body_table(L):
        m = fraction * mass
        at_com = m * np.diag(np.hstack((gyration ** 2, (1, 1, 1))))
        H_fg = np.eye(4)
        H_fg[0:3, 3] = com
        Ad = Hg.adjoint(Hg.inv(H_fg))
        bodies.append(Body(name=prefix + bname,
                           mass=np.dot(Ad.T, np.dot(at_com, Ad))))

    for (parent, offset, joint_class, child) in _link_table(L):
        anchor_body = world.ground if parent is None else bodies[prefix + parent]
        world.add_link(SubFrame(anchor_body, Hg.transl(*offset)), joint_class(),
                       bodies[prefix + child])

    tags = NamedObjectsList()
    for (tname, bname, position) in _tag_table(L, h):
        tag = SubFrame(bodies[prefix + bname], Hg.transl(*position), prefix + tname)
        tags.append(tag)
        world.register(tag)

    for tname in FOOT_POINTS:
        world.register(Point(tags[prefix + tname], name=prefix + tname))

    world.init()
    return None
