"""The package's own HDF5 writer / reader (arboris_python_amd/h5min.py), which stands in for h5py in the observers
(reference: arboris/observers.py:133-289 writes groups of float64 datasets with h5py):

  * the READER parses the file the HDF5 library wrote for the reference's own tests (tests/golden/ref_human36_masses.h5 =
    the reference's tests/human36.h5): every body's mass matrix against the host model;
  * write -> read round trips: nested groups, many links per group (several symbol table nodes), float64 / float32 /
    integers, scalars and empty arrays, unicode-free names as the reference uses;
  * where the HDF5 tools are installed (h5dump of the HDF5 library, not a dependency), the REAL library reads the written
    file: structure and every value;
  * Hdf5Logger's file path: 'w' and 'a' modes through `observers._write_hdf5`.
"""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from arboris_python_amd import h5min


def _sample(seed=0):
    rng = np.random.default_rng(seed)
    data = {"timeline": np.arange(99.) * 0.005,
            "transforms/Arm": rng.normal(size=(99, 4, 4)), "transforms/Hand": rng.normal(size=(99, 4, 4)),
            "gpositions/Shoulder": rng.normal(size=(99, 1)), "gvelocities/Shoulder": rng.normal(size=(99, 1)),
            "sim/run1/model/mass": rng.normal(size=(5, 3, 3)),
            "ints": np.arange(12, dtype=np.int32).reshape(3, 4), "i64": np.array([-1, 2 ** 40], dtype=np.int64),
            "f32": rng.normal(size=7).astype(np.float32), "scalar": np.float64(3.5), "empty": np.zeros((0, 3))}
    for i in range(70):                                 # more links than one symbol table node holds
        data["many/body%02d" % i] = rng.normal(size=(2, 3))
    return data


def test_reads_the_reference_file_written_by_the_hdf5_library():
    from arboris_python_amd import scenes
    r = h5min.read(os.path.join(GOLDEN, "ref_human36_masses.h5"))
    assert len(r) >= 15 and all(k.startswith("masses/") and v.shape == (6, 6) and v.dtype == np.float64 for k, v in r.items())
    w = scenes.human36_world(0)
    bodies = {b.name: b for b in w.getbodies()}
    hit = 0
    for k, M in r.items():
        name = k.split("/", 1)[1]
        if name in bodies:
            assert np.abs(bodies[name].mass - M).max() < 1e-9 * max(1., np.abs(M).max()), name
            hit += 1
    assert hit >= 15


def test_write_read_round_trip(tmp_path):
    data = _sample()
    fn = str(tmp_path / "t.h5")
    h5min.write(fn, data)
    back = h5min.read(fn)
    assert set(back) == set(data)
    for k, v in data.items():
        v = np.asarray(v)
        assert back[k].shape == v.shape and back[k].dtype == v.dtype and np.array_equal(back[k], v), k
    raw = open(fn, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and len(raw) % 8 == 0
    with pytest.raises(h5min.H5Error):
        h5min.write(fn, {"a": np.zeros(2), "a/b": np.zeros(2)})
    with pytest.raises(h5min.H5Error):
        h5min.write(fn, {"c": np.zeros(2, dtype=complex)})
    with pytest.raises(h5min.H5Error):
        (tmp_path / "junk.h5").write_bytes(b"not an hdf5 file at all")
        h5min.read(str(tmp_path / "junk.h5"))


def _h5dump():
    for c in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_h5dump() is None, reason="the HDF5 tools (h5dump) are not installed")
def test_the_hdf5_library_reads_what_h5min_writes(tmp_path):
    data = _sample(1)
    fn = str(tmp_path / "t.h5")
    h5min.write(fn, data)
    tool = _h5dump()
    p = subprocess.run([tool, "-H", fn], capture_output=True, text=True)
    assert p.returncode == 0 and not p.stderr.strip(), p.stderr
    for k, v in data.items():                           # every link, with its type and shape
        name = k.rsplit("/", 1)[-1]
        assert ('DATASET "%s"' % name) in p.stdout, k
    assert p.stdout.count("H5T_IEEE_F64LE") >= 70 and "H5T_IEEE_F32LE" in p.stdout and "H5T_STD_I32LE" in p.stdout
    assert "( 99, 4, 4 ) / ( 99, 4, 4 )" in p.stdout
    assert subprocess.run([tool, fn], capture_output=True, text=True).returncode == 0        # the whole file
    for k in ("transforms/Arm", "sim/run1/model/mass", "ints", "i64", "f32", "many/body69", "scalar"):
        p = subprocess.run([tool, "-d", "/" + k, "-y", "-w", "0", "-m", "%.17g", fn], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        body = p.stdout[p.stdout.index("DATA {") + 6:]
        vals = np.array([float(x) for x in re.findall(r"-?\d[-+0-9.eE]*", body)])
        ref = np.asarray(data[k]).ravel().astype(np.float64)
        assert vals.size == ref.size and np.array_equal(vals, ref), k


def test_logger_file_modes(tmp_path):
    """observers._write_hdf5 (what Hdf5Logger.finish calls): 'w' replaces the file, 'a' keeps the other datasets."""
    from arboris_python_amd.observers import _write_hdf5
    fn = str(tmp_path / "log.h5")
    _write_hdf5(fn, {"sim/timeline": np.arange(4.), "sim/transforms/Arm": np.ones((4, 4, 4))}, "w")
    _write_hdf5(fn, {"again/timeline": np.arange(2.), "sim/timeline": np.arange(4.) + 1}, "a")
    try:
        import h5py
        with h5py.File(fn, "r") as f:
            back = {}
            f.visititems(lambda n, o: back.__setitem__(n, o[()]) if isinstance(o, h5py.Dataset) else None)
    except ImportError:
        back = h5min.read(fn)
    assert set(back) == {"sim/timeline", "sim/transforms/Arm", "again/timeline"}
    assert np.array_equal(back["sim/timeline"], np.arange(4.) + 1) and back["sim/transforms/Arm"].shape == (4, 4, 4)
    _write_hdf5(fn, {"only": np.zeros(3)}, "w")
    assert set(h5min.read(fn)) == {"only"}


def test_hdf5logger_append_probes_the_existing_file_early_and_rewrites_atomically(tmp_path, monkeypatch):
    """Hdf5Logger(mode='a') without h5py: an existing file the built-in reader cannot parse is refused when the logger is
    BUILT (not in finish(), after the simulation has run); a readable one is kept and extended, through a temporary file."""
    import builtins
    from arboris_python_amd import observers, h5min
    real_import = builtins.__import__

    def no_h5py(name, *a, **kw):
        if name == "h5py":
            raise ImportError("h5py hidden for this test")
        return real_import(name, *a, **kw)
    monkeypatch.setattr(builtins, "__import__", no_h5py)
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x02" * 200)              # a superblock version the reader does not know
    with pytest.raises(ValueError, match="cannot be appended"):
        observers.Hdf5Logger(str(bad), mode='a')
    observers.Hdf5Logger(str(bad), mode='w')                            # overwriting is fine
    good = tmp_path / "good.h5"
    h5min.write(str(good), {"old/x": np.arange(3.)})
    observers.Hdf5Logger(str(good), mode='a')                           # readable: accepted
    observers._write_hdf5(str(good), {"new/y": np.ones((2, 2))}, mode="a")
    back = h5min.read(str(good))
    assert np.array_equal(back["old/x"], np.arange(3.)) and np.array_equal(back["new/y"], np.ones((2, 2)))
    assert not (tmp_path / "good.h5.tmp").exists()
