"""Observers: energy, timing and trajectory logging (host side + batched device logs).

Counterpart of arboris/observers.py (EnergyMonitor :14-67, PerfMonitor :70-130,
Hdf5Logger :133-289).  Two levels:

* ``EnergyMonitor`` / ``PerfMonitor`` / ``TrajectoryLogger`` are ``core.Observer``
  subclasses for the single-world ``simulate()`` loop; they read what the device
  left on the world objects.
* ``batched_trajectory`` turns the per-step device logs of
  ``BatchedWorlds.rollout`` (``arb_rollout``: state and energies written by the
  step kernel itself, one launch for the whole horizon) into the same dataset
  layout as ``Hdf5Logger``: ``timeline (nsteps,)``, ``gpositions/<joint>``,
  ``gvelocities/<joint>``, ``transforms/<name> (nsteps,4,4)``.

``Hdf5Logger`` has the reference's constructor and dataset layout.  Names that end in .h5/.hdf5 give real HDF5
files -- through h5py when it is importable, else through the package's own minimal writer (``h5min``: the plain
groups-of-float64-datasets subset the reference uses, readable by the HDF5 library and by h5py) --; any other
name gives an ``.npz`` archive whose keys are the dataset paths.
"""
import time

import numpy as np

from .core import Observer
from .flatten import flatten_world, JT_FREE


def _com_and_mass(mass):
    """Centre of mass as massmatrix.principalframe places it (massmatrix.py:96-99)."""
    m = mass[5, 5]
    if not m > 0.:
        return np.zeros(3), 0.
    rx = mass[0:3, 3:6] / m
    return np.array([rx[2, 1], rx[0, 2], rx[1, 0]]), float(mass[3, 3])


class EnergyMonitor(Observer):
    """Kinetic, potential and mechanical energy at each step (observers.py:14-67).
    Massless bodies are skipped (the reference's ``principalframe`` asserts on them)."""

    def init(self, world, timeline):
        self._world = world
        self.time = []
        self.kinetic_energy = []
        self.potential_energy = []
        self.mechanichal_energy = []
        self._bodies = list(world.ground.iter_descendant_bodies())
        self._com = [_com_and_mass(np.asarray(b.mass, float)) for b in self._bodies]

    def update(self, dt):
        w = self._world
        self.time.append(w.current_time)
        gvel = w.gvel
        Ec = float(np.dot(gvel, np.dot(w.mass, gvel)) / 2.)
        Ep = 0.
        for body, (c, m) in zip(self._bodies, self._com):
            if m == 0.:
                continue
            h = np.dot(np.dot(body.pose, np.hstack((c, 1.)))[0:3], w.up)
            Ep += m * h
        Ep *= 9.81
        self.kinetic_energy.append(Ec)
        self.potential_energy.append(Ep)
        self.mechanichal_energy.append(Ec + Ep)

    def finish(self):
        pass


class PerfMonitor(Observer):
    """Wall-clock time between successive updates (observers.py:70-130)."""

    def __init__(self, log=False):
        self._log = log
        self._last = None
        self._durations = []

    def init(self, world, timeline):
        self._world = world
        self._last = time.perf_counter()

    def update(self, dt):
        now = time.perf_counter()
        self._durations.append(now - self._last)
        self._last = now

    def finish(self):
        pass

    def get_summary(self):
        d = np.array(self._durations)
        if len(d) == 0:
            return "no step recorded"
        return ("total computation time (s): {0}\nmin computation time (s): {1}\n"
                "mean computation time (s): {2}\nmax computation time (s): {3}"
                .format(d.sum(), d.min(), d.mean(), d.max()))


def _write_hdf5(filename, data, mode="w"):
    """``{dataset path: array}`` into an HDF5 file: h5py when importable, else ``h5min`` (mode 'a': the datasets already
    in the file are kept, those of the same path replaced -- ``h5min`` rewrites the file)."""
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(filename, mode) as f:
            for k, v in data.items():
                if k in f:
                    del f[k]
                f[k] = v
        return
    import os
    from . import h5min
    merged = {}
    if mode == "a" and os.path.exists(filename):
        merged.update(h5min.read(filename))          # (datasets only: attributes of the old file are NOT carried over)
    merged.update(data)
    # the rewrite goes to a temporary file first: a failure midway leaves the old file as it was
    tmp = filename + ".tmp"
    h5min.write(tmp, merged)
    os.replace(tmp, filename)


def _probe_appendable(filename):
    """Can ``_write_hdf5(filename, ..., mode='a')`` keep what the file already holds?  Called when the logger is BUILT, so that an
    existing file the package's own reader cannot parse (chunked or compressed datasets, version-2 object headers) is
    refused before the simulation runs instead of in ``finish()``, after it.  (With h5py there is nothing to probe.)"""
    import os
    try:
        import h5py  # noqa: F401
        return
    except ImportError:
        pass
    if os.path.exists(filename):
        from . import h5min
        try:
            h5min.read(filename)
        except Exception as e:
            raise ValueError("%s exists and cannot be appended to without h5py (%s): use mode='w', another file, or install "
                             "h5py.  (Appending through the built-in writer rewrites the file and drops its attributes.)"
                             % (filename, e))


def _save_datasets(filename, data):
    if filename.endswith((".h5", ".hdf5")):
        _write_hdf5(filename, data)
    else:
        np.savez_compressed(filename, **data)


class TrajectoryLogger(Observer):
    """Records the datasets of ``Hdf5Logger`` (observers.py:133-289) in memory.

    ``flat=True``: one transform per body (``Body.pose``); ``flat=False``: one per
    joint (``Joint.pose``) named after the joint's second frame.
    """

    def __init__(self, save_state=False, save_transforms=True, flat=False):
        self._save_state = save_state
        self._save_transforms = save_transforms
        self._flat = flat
        self.data = {}

    def init(self, world, timeline):
        self._world = world
        self._nb_steps = len(timeline) - 1
        self._step = 0
        self.data = {"timeline": np.zeros(self._nb_steps)}
        self._joints = list(world.iterjoints())
        if self._save_state:
            for j in self._joints:
                shape = (4, 4) if np.ndim(j.gpos) == 2 else (j.ndof,)
                self.data["gpositions/%s" % j.name] = np.zeros((self._nb_steps,) + shape)
                self.data["gvelocities/%s" % j.name] = np.zeros((self._nb_steps, j.ndof))
        if self._save_transforms:
            if self._flat:
                self._transforms = {b.name: b for b in world.iterbodies() if b.name is not None}
            else:
                self._transforms = {j.frames[1].name: j for j in self._joints if j.frames[1].name is not None}
            for name in self._transforms:
                self.data["transforms/%s" % name] = np.zeros((self._nb_steps, 4, 4))

    def update(self, dt):
        k = self._step
        self.data["timeline"][k] = self._world.current_time
        if self._save_state:
            for j in self._joints:
                self.data["gpositions/%s" % j.name][k] = j.gpos
                self.data["gvelocities/%s" % j.name][k] = j.gvel
        if self._save_transforms:
            for name, obj in self._transforms.items():
                self.data["transforms/%s" % name][k] = obj.pose
        self._step += 1

    def finish(self):
        pass

    def save(self, filename):
        _save_datasets(filename, self.data)


class Hdf5Logger(TrajectoryLogger):
    """The reference's file-writing observer (observers.py:133-289), same constructor and dataset layout::

        root/timeline (nsteps,)
        root/gpositions/<joint name>   (nsteps,) + gpos.shape      [save_state]
        root/gvelocities/<joint name>  (nsteps, joint.ndof)        [save_state]
        root/transforms/<name>         (nsteps, 4, 4)              [save_transforms]
        root/model/{gvel, gforce, mass, nleffects, admittance}     [save_model]

    ``transforms`` holds one entry per body (``flat=True``, ``Body.pose``) or per joint (``flat=False``,
    ``Joint.pose`` under the name of the joint's second frame), plus the world's moving sub-frames (the contact
    frames).  ``root`` is ``group`` inside the file (default "/").  The file is written by ``finish()``: HDF5
    when the name ends in .h5/.hdf5 (h5py when importable, else the package's own ``h5min`` writer: the same
    groups and float64 datasets, readable by the HDF5 library) or a ``.npz`` archive whose keys are the dataset paths.
    (The reference stores the ``model`` datasets in the ``transforms`` group and reads a ``World.admittance``
    attribute that does not exist; this class follows its documented layout.)
    """

    def __init__(self, filename, group="/", mode='a', save_state=False, save_transforms=True, flat=False,
                 save_model=False):
        TrajectoryLogger.__init__(self, save_state=save_state, save_transforms=save_transforms, flat=flat)
        if mode not in ('a', 'w'):
            raise ValueError("mode must be 'w' or 'a'")
        self._filename, self._mode = filename, mode
        self._group = "/".join(g for g in group.split("/") if g)
        self._save_model = save_model
        self._hdf5 = filename.endswith((".h5", ".hdf5"))
        if self._hdf5 and mode == 'a':
            _probe_appendable(filename)

    @property
    def root(self):
        """path of the group the datasets go to"""
        return "/" + self._group

    def init(self, world, timeline):
        TrajectoryLogger.init(self, world, timeline)
        if self._save_transforms:
            for f in world.itermovingsubframes():          # contact frames (observers.py:239-240)
                if f.name is not None:
                    self._transforms[f.name] = f
                    self.data["transforms/%s" % f.name] = np.zeros((self._nb_steps, 4, 4))
        if self._save_model:
            n = world.ndof
            for name, shape in (("gvel", (n,)), ("gforce", (n,)), ("mass", (n, n)), ("nleffects", (n, n)),
                                ("admittance", (n, n))):
                self.data["model/%s" % name] = np.zeros((self._nb_steps,) + shape)

    def update(self, dt):
        k = self._step
        TrajectoryLogger.update(self, dt)
        if self._save_model:
            w = self._world
            self.data["model/gvel"][k] = w.gvel
            self.data["model/gforce"][k] = w.gforce
            self.data["model/mass"][k] = w.mass
            self.data["model/nleffects"][k] = w.nleffects
            self.data["model/admittance"][k] = w._admittance

    def finish(self):
        prefix = self._group + "/" if self._group else ""
        if self._hdf5:
            _write_hdf5(self._filename, {prefix + key: v for key, v in self.data.items()}, self._mode)
            return
        data = {}
        import os
        if self._mode == 'a' and os.path.exists(self._filename):
            with np.load(self._filename) as old:
                data.update({k: old[k] for k in old.files})
        data.update({prefix + k: v for k, v in self.data.items()})
        np.savez_compressed(self._filename, **data)


def batched_trajectory(bw, world, log, dt, t0=0., world_index=0, flat=True, save_state=True):
    """Hdf5Logger-layout datasets of ONE world of a batched rollout.

    ``bw``: the ``BatchedWorlds`` that produced ``log = bw.rollout(...)``;
    ``world``: the ``core.World`` the model was flattened from (names only).
    Body poses are evaluated on the device from the logged states in one
    ``arb_inspect`` call over the whole horizon.
    """
    import torch
    m = bw.model
    q = log["q"][:, world_index].contiguous()            # (nsteps, nq)
    dq = log["dq"][:, world_index].contiguous()
    nsteps = q.shape[0]
    data = {"timeline": t0 + dt * np.arange(nsteps)}
    joints = list(world.iterjoints())
    qh, dqh = q.cpu().numpy(), dq.cpu().numpy()
    if save_state:
        for b, j in enumerate(joints):
            qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
            ds = slice(int(m.dof_off[b]), int(m.dof_off[b] + m.jnd[b]))
            gp = qh[:, qs].reshape(nsteps, 4, 4) if m.jtype[b] == JT_FREE else qh[:, qs]
            data["gpositions/%s" % j.name] = gp.astype(np.float64)
            data["gvelocities/%s" % j.name] = dqh[:, ds].astype(np.float64)
    if flat:
        poses = bw.inspect(q, dq, float(dt), ["pose"], skip_constraints=True)["pose"].cpu().numpy()
        data["transforms/%s" % world.ground.name] = np.tile(np.eye(4), (nsteps, 1, 1))
        for b, body in enumerate(world.ground.iter_descendant_bodies()):
            if body.name is not None:
                data["transforms/%s" % body.name] = poses[:, b].astype(np.float64)
    else:
        # Joint.pose is the joint-local transform: evaluated from the logged positions with the
        # host joint classes (this is bookkeeping for viewers, not part of the step)
        from . import joints as J
        cls = {0: J.FreeJoint, 1: J.RzRyRxJoint, 2: J.RzRyJoint, 3: J.RzRxJoint, 4: J.RyRxJoint,
               5: J.RzJoint, 6: J.RyJoint, 7: J.RxJoint, 8: J.TxTyTzJoint}
        for b, j in enumerate(joints):
            name = j.frames[1].name
            if name is None:
                continue
            qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
            out = np.zeros((nsteps, 4, 4))
            for k in range(nsteps):
                out[k] = cls[int(m.jtype[b])](gpos=qh[k, qs].astype(np.float64)).pose
            data["transforms/%s" % name] = out
    if "energy" in log:
        e = log["energy"][:, world_index].cpu().numpy().astype(np.float64)
        data["energy/kinetic"] = e[:, 0]
        data["energy/potential"] = e[:, 1]
    return data


def save_trajectory(filename, data):
    """Write the datasets of ``batched_trajectory`` / ``TrajectoryLogger.data``."""
    _save_datasets(filename, data)
