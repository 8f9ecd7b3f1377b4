"""``World.parse``-driven exporters (arboris/core.py:562-606 is the hook; the reference's
consumers are the Collada/OSG drawers of arboris/_visu.py, visu_collada.py:326-388).

``SceneGraphExporter`` turns a world into a plain nested structure (JSON-able) that has the
same hierarchy those drawers build: ground -> frames / shapes / links -> child bodies ..., plus
the constraints and controllers.  Together with ``observers.save_trajectory`` (per-frame
``transforms/<name>`` arrays in the Hdf5Logger layout) it is what an external viewer needs to
replay a device rollout.
"""
import json

import numpy as np

from .core import Body, Frame, Shape, Joint, Constraint, Controller

__all__ = ["ParseRecorder", "SceneGraphExporter", "export_scene"]


def _label(obj):
    return "%s:%s" % (type(obj).__name__, obj.name)


class ParseRecorder(object):
    """Records the sequence of hook calls ``World.parse`` makes (used by the tests to pin the
    traversal order against the reference's)."""

    def __init__(self):
        self.calls = []

    def init_parse(self, ground, up, current_time):
        self.calls.append("init_parse %s" % _label(ground))

    def register(self, obj):
        self.calls.append("register %s" % _label(obj))

    def add_link(self, f0, j, f1):
        self.calls.append("add_link %s %s %s" % (_label(f0), _label(j), _label(f1)))


class SceneGraphExporter(object):
    """``world.parse(SceneGraphExporter())`` -> ``.scene`` (dict)."""

    def __init__(self):
        self.scene = None
        self._nodes = {}

    def init_parse(self, ground, up, current_time):
        self.scene = dict(up=[float(x) for x in up], time=float(current_time), root=None,
                          constraints=[], controllers=[])

    def _node(self, frame):
        if frame not in self._nodes:
            self._nodes[frame] = dict(name=frame.name, kind=type(frame).__name__,
                                      bpose=np.asarray(frame.bpose, float).tolist(),
                                      frames=[], shapes=[], links=[])
        return self._nodes[frame]

    def register(self, obj):
        if isinstance(obj, Body):
            node = self._node(obj)
            node["mass"] = np.asarray(obj.mass, float).tolist()
            if self.scene["root"] is None:
                self.scene["root"] = node
        elif isinstance(obj, Frame):
            self._node(obj.body)["frames"].append(self._node(obj))
        elif isinstance(obj, Shape):
            d = dict(name=obj.name, kind=type(obj).__name__)
            for attr in ("radius", "half_extents", "length", "coeffs"):
                if hasattr(obj, attr):
                    d[attr] = np.asarray(getattr(obj, attr), float).tolist()
            self._node(obj.frame)["shapes"].append(d)
        elif isinstance(obj, Constraint):
            self.scene["constraints"].append(dict(name=obj.name, kind=type(obj).__name__,
                                                  frames=[f.name for f in getattr(obj, "_frames", ())]))
        elif isinstance(obj, Controller):
            self.scene["controllers"].append(dict(name=obj.name, kind=type(obj).__name__))

    def add_link(self, f0, j, f1):
        assert isinstance(j, Joint)
        self._node(f0)["links"].append(dict(joint=j.name, kind=type(j).__name__,
                                            gpos=np.asarray(j.gpos, float).tolist(),
                                            child_frame=f1.name, child=self._node(f1.body)))


def export_scene(world, path=None):
    """Scene graph of ``world`` as a dict; also written to ``path`` as JSON when given."""
    drv = SceneGraphExporter()
    world.parse(drv)
    if path is not None:
        with open(path, "w") as f:
            json.dump(drv.scene, f)
    return drv.scene
