"""COLLADA 1.4.1 export of a world and of a recorded trajectory.

Counterpart of ``write_collada_scene`` / ``write_collada_animation`` of the reference
(arboris/visu_collada.py:326-363) for the build's ``World``: same two entry points and the same
``flat`` switch, written against ``World.parse`` (core.py:562-606).  Unlike the reference it needs
neither the ``scene.dae`` / ``shapes.dae`` templates nor the external ``h5toanim`` tool: geometry
is generated here and the animation is written directly from the trajectory datasets
(``observers.TrajectoryLogger.data`` / ``observers.batched_trajectory``: ``timeline`` and
``transforms/<name>`` arrays of shape (nsteps, 4, 4), the Hdf5Logger layout of
observers.py:155-192), so a rollout computed on the GPU can be replayed in any COLLADA viewer.

Node layout (``<node id=NAME><matrix sid="matrix">``, matrices row-major as COLLADA wants):

* ``flat=True``: every body is a child of the ground node and carries its absolute pose
  ``Body.pose`` -- the transforms a flat trajectory holds (observers.py:229-232);
* ``flat=False``: the kinematic tree.  The node named after a child body sits under the node of the
  joint's parent frame and carries ``Joint.pose`` (what a non-flat trajectory holds, keyed by the
  child body's name, observers.py:233-238); when the joint's second frame is a sub-frame of the
  child body, a constant node ``NAME.origin`` below it brings the children back to the body frame.

Sub-frames are constant nodes under their body; shapes are ``instance_geometry`` of meshes built
here (box, ellipsoid/sphere, cylinder, a marker for points, a finite patch for planes).
"""
import datetime
import xml.etree.ElementTree as ET

import numpy as np

from .core import Body, Frame, Joint, Shape
from . import shapes as _shapes

__all__ = ["write_collada_scene", "write_collada_animation", "ColladaScene"]

NS = "http://www.collada.org/2005/11/COLLADASchema"


def _fmt(values):
    return " ".join(repr(float(x)) for x in np.asarray(values, dtype=float).ravel())


def _safe_id(name):
    return "".join(ch if (ch.isalnum() or ch in "_-.") else "_" for ch in str(name))


# ---------------------------------------------------------------------------
# meshes (vertices (n,3), triangles (m,3)), all centred on the shape's frame
# ---------------------------------------------------------------------------
def _box_mesh(half):
    hx, hy, hz = [float(h) for h in half]
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = [(a, b, c) for (a, b, c, d) in quads] + [(a, c, d) for (a, b, c, d) in quads]
    return v, np.array(t)


def _ellipsoid_mesh(radii, nlat=8, nlon=12):
    rx, ry, rz = [float(r) for r in radii]
    v = [[0., 0., rz]]
    for i in range(1, nlat):
        th = np.pi * i / nlat
        for j in range(nlon):
            ph = 2. * np.pi * j / nlon
            v.append([rx * np.sin(th) * np.cos(ph), ry * np.sin(th) * np.sin(ph), rz * np.cos(th)])
    v.append([0., 0., -rz])
    t = []
    ring = lambda i, j: 1 + (i - 1) * nlon + (j % nlon)
    for j in range(nlon):
        t.append((0, ring(1, j), ring(1, j + 1)))
        t.append((len(v) - 1, ring(nlat - 1, j + 1), ring(nlat - 1, j)))
    for i in range(1, nlat - 1):
        for j in range(nlon):
            t.append((ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)))
            t.append((ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)))
    return np.array(v), np.array(t)


def _cylinder_mesh(length, radius, nseg=16):
    h = 0.5 * float(length)
    v = [[0., 0., -h], [0., 0., h]]
    for j in range(nseg):
        ph = 2. * np.pi * j / nseg
        v.append([radius * np.cos(ph), radius * np.sin(ph), -h])
        v.append([radius * np.cos(ph), radius * np.sin(ph), h])
    t = []
    for j in range(nseg):
        a0, a1 = 2 + 2 * j, 3 + 2 * j
        b0, b1 = 2 + 2 * ((j + 1) % nseg), 3 + 2 * ((j + 1) % nseg)
        t += [(0, b0, a0), (1, a1, b1), (a0, b0, b1), (a0, b1, a1)]
    return np.array(v), np.array(t)


def _plane_mesh(coeffs, size=1.):
    """A square patch of the plane n.x + d = 0, centred on the point of the plane closest to the origin."""
    n = np.asarray(coeffs[0:3], float)
    d = float(coeffs[3])
    n = n / np.linalg.norm(n)
    a = np.eye(3)[int(np.argmin(np.abs(n)))]
    u = np.cross(n, a); u /= np.linalg.norm(u)
    w = np.cross(n, u)
    c = -d * n
    v = np.array([c + size * (su * u + sw * w) for (su, sw) in ((-1, -1), (1, -1), (1, 1), (-1, 1))])
    return v, np.array([(0, 1, 2), (0, 2, 3)])


def _shape_mesh(shape, point_size=0.01):
    if isinstance(shape, _shapes.Box):
        return _box_mesh(shape.half_extents)
    if isinstance(shape, _shapes.Sphere):
        return _ellipsoid_mesh((shape.radius,) * 3)
    if isinstance(shape, _shapes.Cylinder):
        return _cylinder_mesh(shape.length, shape.radius)
    if isinstance(shape, _shapes.Plane):
        return _plane_mesh(shape.coeffs)
    if isinstance(shape, _shapes.Point):
        return _ellipsoid_mesh((point_size,) * 3, nlat=2, nlon=4)        # an octahedron
    if hasattr(shape, "radii"):
        return _ellipsoid_mesh(shape.radii)
    raise NotImplementedError("no COLLADA geometry for %s" % type(shape).__name__)


# ---------------------------------------------------------------------------
class ColladaScene(object):
    """``world.parse`` target building the COLLADA document (``.tree`` once ``finish()`` ran)."""

    def __init__(self, flat=False, scale=1., color=(0.6, 0.6, 0.7, 1.0)):
        self.flat = bool(flat)
        self.scale = float(scale)
        self.color = color
        self.root = ET.Element("COLLADA", {"xmlns": NS, "version": "1.4.1"})
        asset = ET.SubElement(self.root, "asset")
        now = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ")
        ET.SubElement(asset, "created").text = now
        ET.SubElement(asset, "modified").text = now
        ET.SubElement(asset, "unit", {"name": "meter", "meter": repr(self.scale)})
        self._up = ET.SubElement(asset, "up_axis")
        self._effects = ET.SubElement(self.root, "library_effects")
        self._materials = ET.SubElement(self.root, "library_materials")
        self._geoms = ET.SubElement(self.root, "library_geometries")
        scenes = ET.SubElement(self.root, "library_visual_scenes")
        self._scene = ET.SubElement(scenes, "visual_scene", {"id": "myscene"})
        inst = ET.SubElement(ET.SubElement(self.root, "scene"), "instance_visual_scene", {"url": "#myscene"})
        del inst
        self._nodes = {}          # Frame -> xml node that carries its children
        self._ids = set()
        self._ground = None
        self.animated = []        # ids of the nodes a trajectory moves
        self._material()

    # -- helpers -----------------------------------------------------------
    def _material(self):
        eff = ET.SubElement(self._effects, "effect", {"id": "arb_effect"})
        tech = ET.SubElement(ET.SubElement(eff, "profile_COMMON"), "technique", {"sid": "common"})
        ET.SubElement(ET.SubElement(ET.SubElement(tech, "phong"), "diffuse"), "color").text = _fmt(self.color)
        mat = ET.SubElement(self._materials, "material", {"id": "arb_material"})
        ET.SubElement(mat, "instance_effect", {"url": "#arb_effect"})

    def _unique(self, name):
        base = _safe_id(name)
        uid, k = base, 1
        while uid in self._ids:
            k += 1
            uid = "%s.%d" % (base, k)
        self._ids.add(uid)
        return uid

    def _new_node(self, parent, name, pose):
        uid = self._unique(name)
        node = ET.SubElement(parent, "node", {"id": uid, "name": str(name)})
        ET.SubElement(node, "matrix", {"sid": "matrix"}).text = _fmt(pose)
        return node

    def _geometry(self, shape):
        v, t = _shape_mesh(shape)
        gid = self._unique("geom." + str(shape.name))
        mesh = ET.SubElement(ET.SubElement(self._geoms, "geometry", {"id": gid}), "mesh")
        src = ET.SubElement(mesh, "source", {"id": gid + ".pos"})
        ET.SubElement(src, "float_array", {"id": gid + ".pos.arr", "count": str(v.size)}).text = _fmt(v)
        acc = ET.SubElement(ET.SubElement(src, "technique_common"), "accessor",
                            {"source": "#" + gid + ".pos.arr", "count": str(len(v)), "stride": "3"})
        for ax in "XYZ":
            ET.SubElement(acc, "param", {"name": ax, "type": "float"})
        vert = ET.SubElement(mesh, "vertices", {"id": gid + ".vtx"})
        ET.SubElement(vert, "input", {"semantic": "POSITION", "source": "#" + gid + ".pos"})
        tri = ET.SubElement(mesh, "triangles", {"count": str(len(t)), "material": "mat"})
        ET.SubElement(tri, "input", {"semantic": "VERTEX", "source": "#" + gid + ".vtx", "offset": "0"})
        ET.SubElement(tri, "p").text = " ".join(str(int(i)) for i in t.ravel())
        return gid

    # -- World.parse hooks ---------------------------------------------------
    def init_parse(self, ground, up, current_time):
        up = np.asarray(up, float).ravel()
        self._up.text = ("X_UP", "Y_UP", "Z_UP")[int(np.argmax(np.abs(up)))]
        self._ground = self._new_node(self._scene, ground.name or "ground", np.eye(4))
        self._nodes[ground] = self._ground

    def add_link(self, f0, joint, f1):
        assert isinstance(joint, Joint)
        body = f1.body
        if self.flat:
            node = self._new_node(self._ground, body.name, body.pose)
            self._nodes[body] = node
        else:
            parent = self._nodes[f0]
            node = self._new_node(parent, body.name, joint.pose)
            if f1 is body:
                self._nodes[body] = node
            else:                        # the joint ends on a sub-frame of the child body
                inv = np.linalg.inv(np.asarray(f1.bpose, float))
                self._nodes[body] = self._new_node(node, "%s.origin" % body.name, inv)
        self.animated.append(node.get("id"))

    def register(self, obj):
        if isinstance(obj, Body):
            return                       # created by init_parse / add_link
        if isinstance(obj, Frame):
            if obj not in self._nodes:
                self._nodes[obj] = self._new_node(self._nodes[obj.body], obj.name, obj.bpose)
        elif isinstance(obj, Shape):
            frame = obj.frame
            if frame not in self._nodes:     # a shape on a sub-frame parsed before its registration
                self._nodes[frame] = self._new_node(self._nodes[frame.body], frame.name, frame.bpose)
            gid = self._geometry(obj)
            ig = ET.SubElement(self._nodes[frame], "instance_geometry", {"url": "#" + gid})
            tc = ET.SubElement(ET.SubElement(ig, "bind_material"), "technique_common")
            ET.SubElement(tc, "instance_material", {"symbol": "mat", "target": "#arb_material"})
        # constraints and controllers have no visual representation

    def finish(self):
        self.tree = ET.ElementTree(self.root)
        return self.tree


def _indent(elem, level=0):
    pad = "\n" + "\t" * level
    if len(elem):
        if not (elem.text or "").strip():
            elem.text = pad + "\t"
        for child in elem:
            _indent(child, level + 1)
            if not (child.tail or "").strip():
                child.tail = pad + "\t"
        child.tail = pad
    if level and not (elem.tail or "").strip():
        elem.tail = pad


def write_collada_scene(world, dae_filename, flat=False, scale=1.):
    """Write the visual description of ``world`` (its current configuration: call
    ``world.update_geometric()`` first) to ``dae_filename``.  Returns the ``ColladaScene``."""
    drv = ColladaScene(flat=flat, scale=scale)
    world.parse(drv)
    tree = drv.finish()
    _indent(tree.getroot())
    tree.write(dae_filename, encoding="utf-8", xml_declaration=True)
    return drv


def _read_hdf5(filename):
    try:
        import h5py
    except ImportError:
        from . import h5min
        return h5min.read(filename)
    out = {}
    with h5py.File(filename, "r") as f:
        f.visititems(lambda n, o: out.__setitem__(n, o[()]) if isinstance(o, h5py.Dataset) else None)
    return out


def write_collada_animation(collada_animation, collada_scene, trajectory, hdf5_group="/", *, prefix="transforms/"):
    """Add one ``<animation>`` per ``transforms/<name>`` dataset of ``trajectory`` to the scene file
    ``collada_scene`` and write the result to ``collada_animation``.

    ``trajectory``: the path of an HDF5 file written by ``Hdf5Logger`` -- the reference's signature
    ``(collada_animation, collada_scene, hdf5_file, hdf5_group="/")``, visu_collada.py:343-364, which hands the job to the
    external ``h5toanim`` program; read here with h5py or the package's own reader --, or a mapping with ``timeline``
    (nsteps,) and ``transforms/<name>`` (nsteps, 4, 4) (``TrajectoryLogger.data``, ``batched_trajectory(...)``), or
    the path of an ``.npz`` written by ``observers.save_trajectory``.  ``hdf5_group``: the group of the file (or the
    key prefix of the archive, or of the mapping) the logger wrote to.  ``prefix`` is keyword-only: the fourth positional
    argument is the reference's ``hdf5_group``.  ``<name>`` must be the id of a node of the scene; whether the
    matrices are absolute poses or joint poses must match the ``flat`` flag the scene was written with.
    """
    if isinstance(trajectory, str):
        if trajectory.endswith((".h5", ".hdf5")):
            trajectory = _read_hdf5(trajectory)
        elif trajectory.endswith(".npz"):
            trajectory = dict(np.load(trajectory))
        else:
            raise ValueError("trajectory file %r: expected .h5 / .hdf5 (Hdf5Logger) or .npz (observers.save_trajectory)" % trajectory)
    group = "/".join(g for g in hdf5_group.split("/") if g)
    if group:                                      # (files, archives and mappings alike)
        trajectory = {k[len(group) + 1:]: v for k, v in trajectory.items() if k.startswith(group + "/")}
    ET.register_namespace("", NS)
    tree = ET.parse(collada_scene)
    root = tree.getroot()
    q = lambda tag: "{%s}%s" % (NS, tag)
    node_ids = set(n.get("id") for n in root.iter(q("node")))
    t = np.asarray(trajectory["timeline"], float).ravel()
    lib = root.find(q("library_animations"))
    if lib is None:
        lib = ET.Element(q("library_animations"))
        root.insert(list(root).index(root.find(q("library_visual_scenes"))), lib)
    count = 0
    for key in sorted(k for k in trajectory.keys() if k.startswith(prefix)):
        name = _safe_id(key[len(prefix):])
        if name not in node_ids:
            raise KeyError("the scene has no node %r to animate" % name)
        H = np.asarray(trajectory[key], float)
        n = min(len(t), H.shape[0])
        assert H.shape[1:] == (4, 4)
        aid = name + ".anim"
        anim = ET.SubElement(lib, q("animation"), {"id": aid})

        def source(suffix, data, stride, params):
            src = ET.SubElement(anim, q("source"), {"id": aid + suffix})
            if params[0][1] == "name":
                arr = ET.SubElement(src, q("Name_array"), {"id": aid + suffix + ".arr", "count": str(len(data))})
                arr.text = " ".join(data)
            else:
                arr = ET.SubElement(src, q("float_array"), {"id": aid + suffix + ".arr", "count": str(np.size(data))})
                arr.text = _fmt(data)
            acc = ET.SubElement(ET.SubElement(src, q("technique_common")), q("accessor"),
                                {"source": "#" + aid + suffix + ".arr", "count": str(n), "stride": str(stride)})
            for pname, ptype in params:
                ET.SubElement(acc, q("param"), {"name": pname, "type": ptype})

        source(".input", t[:n], 1, [("TIME", "float")])
        source(".output", H[:n], 16, [("TRANSFORM", "float4x4")])
        source(".interp", ["LINEAR"] * n, 1, [("INTERPOLATION", "name")])
        smp = ET.SubElement(anim, q("sampler"), {"id": aid + ".sampler"})
        for sem, suffix in (("INPUT", ".input"), ("OUTPUT", ".output"), ("INTERPOLATION", ".interp")):
            ET.SubElement(smp, q("input"), {"semantic": sem, "source": "#" + aid + suffix})
        ET.SubElement(anim, q("channel"), {"source": "#" + aid + ".sampler", "target": name + "/matrix"})
        count += 1
    _indent(root)
    tree.write(collada_animation, encoding="utf-8", xml_declaration=True)
    return count
