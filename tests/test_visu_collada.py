"""COLLADA export (arboris_python_amd/visu_collada.py), counterpart of the reference's
tests/test_visu_collada.py: scene of the simplearm with shapes in both layouts, animation from the
trajectories of the reference's own golden files (tests/simplearm_flat.h5 / simplearm_notflat.h5
payloads, kept in tests/golden/g1_simplearm.npz).  Host only: joint poses come from the host joint
classes, body poses of the flat layout are set from the golden data."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from conftest import load_golden
from arboris_python_amd.core import World
from arboris_python_amd.controllers import WeightController
from arboris_python_amd.robots.simplearm import add_simplearm
from arboris_python_amd.visu_collada import NS, write_collada_animation, write_collada_scene

Q = lambda tag: "{%s}%s" % (NS, tag)
ORDER = ("Hand", "Arm", "Forearm")            # dataset order of the golden files


def make_world():
    w = World()
    w.register(WeightController())
    add_simplearm(w, with_shapes=True)
    w.getjoints()["Shoulder"].gpos[0] = 3.14 / 4
    return w


def matrix_of(node):
    return np.array([float(x) for x in node.find(Q("matrix")).text.split()]).reshape(4, 4)


def test_scene_not_flat(tmp_path):
    w = make_world()
    f = str(tmp_path / "scene.dae")
    drv = write_collada_scene(w, f, flat=False)
    root = ET.parse(f).getroot()
    assert root.tag == Q("COLLADA") and root.get("version") == "1.4.1"
    assert root.find(Q("asset")).find(Q("up_axis")).text == "Y_UP"            # World.up = (0, 1, 0)
    nodes = {n.get("id"): n for n in root.iter(Q("node"))}
    for j in w.getjoints():
        body = j.frames[1].body
        assert np.allclose(matrix_of(nodes[body.name]), np.asarray(j.pose), atol=1e-15)
    # kinematic chain ground -> Arm -> ... -> Forearm -> ... -> Hand
    parent = {c.get("id"): p.get("id") for p in root.iter(Q("node")) for c in p.findall(Q("node"))}

    def ancestors(i):
        out = []
        while i in parent:
            i = parent[i]
            out.append(i)
        return out
    assert "Arm" in ancestors("Forearm") and "Forearm" in ancestors("Hand") and "ground" in ancestors("Arm")
    assert sorted(drv.animated) == ["Arm", "Forearm", "Hand"]
    # one mesh per registered shape, each instantiated once and well formed
    geoms = root.find(Q("library_geometries")).findall(Q("geometry"))
    assert len(geoms) == len(w.getshapes()) > 0
    assert len(list(root.iter(Q("instance_geometry")))) == len(geoms)
    for g in geoms:
        tri = g.find(Q("mesh")).find(Q("triangles"))
        idx = [int(x) for x in tri.find(Q("p")).text.split()]
        nv = int(g.find(Q("mesh")).find(Q("source")).find(Q("technique_common")).find(Q("accessor")).get("count"))
        assert len(idx) == 3 * int(tri.get("count")) and 0 <= min(idx) and max(idx) < nv


def test_scene_flat_uses_body_poses(tmp_path):
    g = load_golden("g1_simplearm.npz")
    w = make_world()
    poses = dict(zip(ORDER, g["h5_flat_HandArmForearm"][:, 0]))
    for b in w.iterbodies():
        if b.name in poses:
            b._pose = poses[b.name]                 # what update_geometric() leaves (needs the GPU)
    f = str(tmp_path / "flat.dae")
    write_collada_scene(w, f, flat=True)
    root = ET.parse(f).getroot()
    ground = [n for n in root.iter(Q("node")) if n.get("id") == "ground"][0]
    children = {n.get("id"): n for n in ground.findall(Q("node"))}
    for name in ORDER:
        assert np.allclose(matrix_of(children[name]), poses[name], atol=1e-15)


@pytest.mark.parametrize("flat", [True, False])
def test_animation_from_reference_trajectories(tmp_path, flat):
    g = load_golden("g1_simplearm.npz")
    key = "h5_flat" if flat else "h5_notflat"
    traj = {"timeline": g[key + "_timeline"]}
    for name, H in zip(ORDER, g[key + "_HandArmForearm"]):
        traj["transforms/" + name] = H
    w = make_world()
    if flat:
        for b in w.iterbodies():
            if b.name in ORDER:
                b._pose = traj["transforms/" + b.name][0]
    scene, anim = str(tmp_path / "s.dae"), str(tmp_path / "a.dae")
    write_collada_scene(w, scene, flat=flat)
    assert write_collada_animation(anim, scene, traj) == 3
    root = ET.parse(anim).getroot()
    lib = root.find(Q("library_animations"))
    assert [c.tag for c in root].index(Q("library_animations")) < [c.tag for c in root].index(Q("library_visual_scenes"))
    anims = {a.get("id"): a for a in lib.findall(Q("animation"))}
    assert sorted(anims) == ["Arm.anim", "Forearm.anim", "Hand.anim"]
    for name in ORDER:
        a = anims[name + ".anim"]
        arrays = {s.get("id"): s for s in a.findall(Q("source"))}
        t = np.array([float(x) for x in arrays[name + ".anim.input"].find(Q("float_array")).text.split()])
        H = np.array([float(x) for x in arrays[name + ".anim.output"].find(Q("float_array")).text.split()])
        assert np.array_equal(t, traj["timeline"])
        assert np.array_equal(H.reshape(-1, 4, 4), traj["transforms/" + name])
        assert a.find(Q("channel")).get("target") == name + "/matrix"
        assert arrays[name + ".anim.interp"].find(Q("Name_array")).text.split() == ["LINEAR"] * len(t)
    # the first sample is the configuration the scene was written in
    nodes = {n.get("id"): n for n in root.iter(Q("node"))}
    for name in ORDER:
        assert np.allclose(matrix_of(nodes[name]), traj["transforms/" + name][0], atol=1e-12)


def test_animation_from_an_hdf5_file_and_group(tmp_path):
    """The reference's call, write_collada_animation(anim, scene, hdf5_file, hdf5_group) (visu_collada.py:343-364): the
    trajectory comes from an HDF5 file as Hdf5Logger writes it (here through the package's own writer), inside a group."""
    from arboris_python_amd.observers import _write_hdf5
    g = load_golden("g1_simplearm.npz")
    data = {"sim/run/timeline": g["h5_flat_timeline"]}
    for name, H in zip(ORDER, g["h5_flat_HandArmForearm"]):
        data["sim/run/transforms/" + name] = H
    data["other/timeline"] = np.zeros(3)
    h5 = str(tmp_path / "log.h5")
    _write_hdf5(h5, data, "w")
    w = make_world()
    for b in w.iterbodies():
        if b.name in ORDER:
            b._pose = data["sim/run/transforms/" + b.name][0]
    scene, anim = str(tmp_path / "s.dae"), str(tmp_path / "a.dae")
    write_collada_scene(w, scene, flat=True)
    assert write_collada_animation(anim, scene, h5, "/sim/run") == 3
    root = ET.parse(anim).getroot()
    a = {x.get("id"): x for x in root.find(Q("library_animations")).findall(Q("animation"))}["Arm.anim"]
    arrays = {s_.get("id"): s_ for s_ in a.findall(Q("source"))}
    H = np.array([float(x) for x in arrays["Arm.anim.output"].find(Q("float_array")).text.split()])
    assert np.array_equal(H.reshape(-1, 4, 4), data["sim/run/transforms/Arm"])


def test_animation_rejects_unknown_nodes(tmp_path):
    w = make_world()
    scene = str(tmp_path / "s.dae")
    write_collada_scene(w, scene, flat=False)
    with pytest.raises(KeyError):
        write_collada_animation(str(tmp_path / "a.dae"), scene,
                                {"timeline": np.zeros(2), "transforms/NoSuchBody": np.tile(np.eye(4), (2, 1, 1))})
