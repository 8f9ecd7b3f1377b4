"""Batched observers: device energy monitor and trajectory logs."""
import numpy as np
import pytest

import arb_oracle as O
from conftest import load_golden, load_model

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from arboris_python_amd.batch import BatchedWorlds  # noqa: E402
from arboris_python_amd import synth, scenes, observers  # noqa: E402
from arboris_python_amd.core import simulate  # noqa: E402


def oracle_energy(m, q, dq):
    """EnergyMonitor.update (observers.py:40-51) from the oracle's M and body poses."""
    d = O.update_dynamic(m, q, dq)
    ke = 0.5 * np.einsum('bi,bij,bj->b', dq, d["M"], dq)
    pe = np.zeros(q.shape[0])
    for b in range(m.nb):
        mass = m.mass[b]
        if not mass[5, 5] > 0:
            continue
        rx = mass[0:3, 3:6] / mass[5, 5]
        c = np.array([rx[2, 1], rx[0, 2], rx[1, 0], 1.])
        pe += mass[3, 3] * (d["pose"][:, b] @ c)[:, 0:3] @ m.up
    return ke, 9.81 * pe


@pytest.mark.parametrize("name,gen,kw", [
    ("human36_c4", synth.standing_states, dict(seed=1, drop=0.03, vel=0.5)),
    ("human36_g", synth.random_states, dict(seed=2)),
    ("snake64_g", synth.random_states, dict(seed=3, angle=0.5, vel=1.0)),
])
def test_device_energy_matches_oracle(name, gen, kw):
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    q, dq = gen(m, 32, **kw)
    ke, pe = oracle_energy(m, q, dq)
    for dtype, tol in ((torch.float64, 1e-11), (torch.float32, 2e-6)):
        tq, tdq = bw.to_device(q, dq, dtype)
        e = bw.inspect(tq, tdq, 5e-3, ["energy"], skip_constraints=True)["energy"].cpu().numpy()
        assert np.max(np.abs(e[:, 0] - ke) / np.maximum(1., np.abs(ke))) < tol
        assert np.max(np.abs(e[:, 1] - pe) / np.maximum(1., np.abs(pe))) < tol
    bw.close()


def test_rollout_logs_equal_stepwise_states():
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    q, dq = synth.standing_states(m, 64, seed=5, drop=0.02, vel=0.2)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(64, torch.float32)
    log = bw.rollout(tq, tdq, 5e-3, 16, cforce=cf)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(64, torch.float32)
    for k in range(16):
        assert torch.equal(log["q"][k], sq) and torch.equal(log["dq"][k], sdq)
        e = bw.inspect(sq, sdq, 5e-3, ["energy"], cforce=scf)["energy"]
        # (two instantiations of the kernel -- production with logs, inspect: the same float64 sums, but the
        # compiler is free to contract their multiply-adds differently: equal to float32 rounding, not bitwise)
        assert torch.allclose(log["energy"][k], e, rtol=2e-6, atol=1e-6)
        bw.step(sq, sdq, 5e-3, 1, cforce=scf)
    torch.cuda.synchronize()
    assert torch.equal(tq, sq) and torch.equal(tdq, sdq)
    bw.close()


def test_energy_drift_series_from_device_energy():
    """tests/test_energy_drift.py + energy_drift.h5: the kinetic-energy series comes from
    the device energy monitor (frozen-hinge quirk emulated between steps)."""
    g = load_golden("g5_energy.npz")
    m, q0, dq0 = load_model("snake9_free_g")
    bw = BatchedWorlds(m)
    tq, tdq = bw.to_device(q0[None], dq0[None], torch.float64)
    tl = g["timeline"]
    frozen_q = [int(m.q_off[b]) for b in g["frozen_bodies"]]
    ke = []
    t = tl[0]
    for tn in tl[1:]:
        dt = float(tn - t)
        log = bw.rollout(tq, tdq, dt, 1, log_state=False)
        ke.append(float(log["energy"][0, 0, 0]))
        tq[0, frozen_q] = 0.
        t += dt
    assert np.max(np.abs(np.array(ke) / g["h5_kinetic_energy"] - 1)) < 1e-7
    bw.close()


@pytest.mark.parametrize("flat", [True, False])
def test_batched_trajectory_matches_reference_h5_layout(flat):
    """Config 1 through the batched logger: datasets named and shaped like Hdf5Logger's,
    values equal to the payload of the reference's simplearm_flat.h5 / simplearm_notflat.h5."""
    g = load_golden("g1_simplearm.npz")
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.robots.simplearm import add_simplearm
    from arboris_python_amd.core import World
    from arboris_python_amd.flatten import flatten_world
    w = World()
    w.register(WeightController())
    add_simplearm(w, with_shapes=True)
    w.getjoints()['Shoulder'].gpos[0] = 3.14 / 4
    m, q0, dq0 = flatten_world(w)
    bw = BatchedWorlds(m)
    # 8 copies of the world: the logger extracts one of them
    tq, tdq = bw.to_device(np.tile(q0, (8, 1)), np.tile(dq0, (8, 1)), torch.float64)
    log = bw.rollout(tq, tdq, 0.01, 99)
    data = observers.batched_trajectory(bw, w, log, 0.01, t0=0., world_index=5, flat=flat)
    assert np.abs(data["timeline"] - g["h5_flat_timeline"]).max() < 1e-12
    h5 = g["h5_flat_HandArmForearm" if flat else "h5_notflat_HandArmForearm"]
    for k, name in enumerate(("Hand", "Arm", "Forearm")):
        assert data["transforms/%s" % name].shape == (99, 4, 4)
        assert np.abs(data["transforms/%s" % name] - h5[k]).max() < 1e-9
    assert data["gpositions/Shoulder"].shape == (99, 1) and data["gvelocities/Wrist"].shape == (99, 1)
    assert data["energy/kinetic"].shape == (99,)
    bw.close()


def test_object_api_observers(tmp_path):
    """EnergyMonitor / TrajectoryLogger / PerfMonitor in the single-world simulate() loop."""
    from arboris_python_amd.controllers import WeightController
    w = scenes.simplearm_world()
    w.getjoints()['Shoulder'].gpos[0] = 3.14 / 4
    nrj, traj, perf = observers.EnergyMonitor(), observers.TrajectoryLogger(save_state=True, flat=True), observers.PerfMonitor()
    simulate(w, np.arange(0, 0.1, 0.01), (nrj, traj, perf))
    g = load_golden("g1_simplearm.npz")
    assert np.abs(traj.data["transforms/Arm"] - g["h5_flat_HandArmForearm"][1][:9]).max() < 1e-9
    assert len(nrj.kinetic_energy) == 9 and nrj.kinetic_energy[0] == 0.
    m, q0, dq0 = load_model("simplearm_g")
    ke, pe = oracle_energy(m, g["traj_q"][:9], g["traj_dq"][:9])
    assert np.abs(np.array(nrj.kinetic_energy) - ke).max() < 1e-10
    assert np.abs(np.array(nrj.potential_energy) - pe).max() < 1e-10
    traj.save(str(tmp_path / "traj.npz"))
    back = np.load(str(tmp_path / "traj.npz"))
    assert back["gpositions/Shoulder"].shape == (9, 1)
    assert "mean computation time" in perf.get_summary()


@pytest.mark.parametrize("ext", ["npz", "h5"])
def test_hdf5logger_layout_and_values(tmp_path, ext):
    """Hdf5Logger through simulate(): the reference's dataset layout (observers.py:155-192), values against
    the reference's simplearm run (tests/golden/g1_simplearm.npz = the payload of simplearm_flat.h5).  As an .npz
    archive and as a real HDF5 file (h5py when installed, else the package's own writer, arboris_python_amd/h5min.py --
    read back here with h5min's reader, by the HDF5 library itself in tests/test_h5min.py)."""
    from conftest import load_golden
    from arboris_python_amd import h5min
    load = (lambda fn: np.load(fn)) if ext == "npz" else h5min.read
    files = (lambda d: d.files) if ext == "npz" else (lambda d: list(d))
    from arboris_python_amd import scenes
    from arboris_python_amd.core import simulate
    from arboris_python_amd.observers import Hdf5Logger
    g = load_golden("g1_simplearm.npz")
    w = scenes.simplearm_world()
    w.getjoints()['Shoulder'].gpos[0] = 3.14 / 4
    timeline = np.arange(0, 1, .01)
    f = str(tmp_path / ("run." + ext))
    simulate(w, timeline, [Hdf5Logger(f, group="sim", mode='w', save_state=True, flat=True, save_model=True)])
    d = load(f)
    n = len(timeline) - 1
    assert d["sim/timeline"].shape == (n,) and np.allclose(d["sim/timeline"], timeline[:-1])
    for j in ("Shoulder", "Elbow", "Wrist"):
        assert d["sim/gpositions/" + j].shape == (n, 1) and d["sim/gvelocities/" + j].shape == (n, 1)
    for k, name in enumerate(("Hand", "Arm", "Forearm")):
        assert np.abs(d["sim/transforms/" + name] - g["h5_flat_HandArmForearm"][k]).max() < 1e-9
    assert d["sim/transforms/ground"].shape == (n, 4, 4)
    for name, shape in (("gvel", (n, 3)), ("gforce", (n, 3)), ("mass", (n, 3, 3)), ("nleffects", (n, 3, 3)),
                        ("admittance", (n, 3, 3))):
        assert d["sim/model/" + name].shape == shape
    assert np.abs(d["sim/gpositions/Shoulder"][:, 0] - g["traj_q"][:, 0]).max() < 1e-9
    # mass . admittance = (M/dt + N)^-1 M ... at least: admittance is the inverse of the impedance M/dt + B + N
    k = 50
    Z = d["sim/model/mass"][k] / 0.01 + d["sim/model/nleffects"][k]
    assert np.abs(Z @ d["sim/model/admittance"][k] - np.eye(3)).max() < 1e-8
    # append mode keeps what is in the file
    simulate(w, timeline[:5], [Hdf5Logger(f, group="again", mode='a', save_transforms=False, save_state=True)])
    d = load(f)
    assert "sim/timeline" in files(d) and d["again/gvelocities/Elbow"].shape == (4, 1)
    assert d["sim/transforms/Arm"].shape == (n, 4, 4)


@pytest.mark.parametrize("name", ["simplearm", "snake9_free"])
def test_device_energy_matches_reference_energy_monitor(name):
    """Kinetic AND potential energy of the device's energy monitor against EnergyMonitor.update executed on the
    reference's objects with the reference's `principalframe` (tests/golden/g13_energy_monitor.npz;
    observers.py:36-51).  float64 1e-10, float32 2e-6."""
    g = load_golden("g13_energy_monitor.npz")
    m, _, _ = load_model("energy_" + name)
    bw = BatchedWorlds(m)
    q, dq = g[name + "_q"], g[name + "_dq"]
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-6)):
        tq, tdq = bw.to_device(q, dq, dtype)
        e = bw.inspect(tq, tdq, 5e-3, ["energy"], skip_constraints=True)["energy"].cpu().numpy()
        assert np.max(np.abs(e[:, 0] - g[name + "_ke"]) / np.maximum(1., np.abs(g[name + "_ke"]))) < tol
        assert np.max(np.abs(e[:, 1] - g[name + "_pe"]) / np.maximum(1., np.abs(g[name + "_pe"]))) < tol
    bw.close()
