"""Device evaluation of the four World step methods for ONE world.

``core.World.update_dynamic / update_controllers / update_constraints /
integrate`` call ``SingleWorldEngine.run``: the world is flattened, evaluated on
the GPU in float64 as a batch of one through ``arb_inspect`` / ``arb_step`` and the
results are scattered back onto the objects exactly where the reference leaves
them (core.py:1272-1288 body attributes, :722-734 world matrices, :812-818
impedance/admittance, :910-937 constraint state, :974-980 joint state).

Two attributes are derived on the host from device results because the device
never forms them: ``Body.nleffects`` (from the device twist, core.py:1276-1288)
and ``World._admittance`` (inverse of the device impedance; the kernels solve
with Z instead of inverting it).
"""
import numpy as np

from . import _capi
from .batch import BatchedWorlds
from .flatten import (flatten_world, JT_FREE, CT_SOFTFINGER, CT_JOINTLIMITS,
                      CT_BALLSOCKET)

_NDOL = {CT_SOFTFINGER: 4, CT_JOINTLIMITS: 1, CT_BALLSOCKET: 3}


def _same_model(a, b):
    da, db = a.to_npz_dict(), b.to_npz_dict()
    if da.keys() != db.keys():
        return False
    for k in da:
        x, y = da[k], db[k]
        if x.shape != y.shape or not np.array_equal(x, y):
            return False
    return True


def _hinv(H):
    out = np.eye(4)
    out[0:3, 0:3] = H[0:3, 0:3].T
    out[0:3, 3] = -H[0:3, 0:3].T @ H[0:3, 3]
    return out


class SingleWorldEngine(object):
    def __init__(self, world, device=0):
        self._device = device
        self._bw = None
        self._warm = None

    def _prepare(self, world):
        import torch
        m, q, dq = flatten_world(world)
        if self._bw is None or not _same_model(self._bw.model, m):
            if self._bw is not None:
                self._bw.close()
            self._bw = BatchedWorlds(m, self._device)
        bw = self._bw
        tq, tdq = bw.to_device(q[None], dq[None], torch.float64)
        cf = np.zeros((1, m.nc, _capi.ARB_MAXDOL))
        for c, con in enumerate(world._constraints):
            f = np.asarray(con._force, float).ravel()
            cf[0, c, :len(f)] = f
        return m, bw, tq, tdq, cf

    def run(self, world, stage, dt):
        import torch
        m, bw, tq, tdq, cf = self._prepare(world)
        bodies = list(world.ground.iter_descendant_bodies())
        n = m.ndof
        tcf = torch.as_tensor(cf, dtype=torch.float64, device=bw.device).contiguous() if m.nc else None
        if stage in ("geometric", "dynamic"):
            want = ["pose"] if stage == "geometric" else ["pose", "twist", "jac", "djac", "M", "B", "N"]
            r = bw.inspect(tq, tdq, 1.0, want, skip_constraints=True)
            r = {k: v.cpu().numpy()[0] for k, v in r.items()}
            world.ground._pose = np.eye(4)
            for b, body in enumerate(bodies):
                body._pose = r["pose"][b].copy()
            if stage == "dynamic":
                world.ground._jacobian = np.zeros((6, n))
                world.ground._djacobian = np.zeros((6, n))
                world.ground._twist = np.zeros(6)
                world.ground._nleffects = np.zeros((6, 6))
                for b, body in enumerate(bodies):
                    body._jacobian = r["jac"][b].copy()
                    body._djacobian = r["djac"][b].copy()
                    body._twist = r["twist"][b].copy()
                    body._nleffects = _body_nleffects(body._twist, m.mass[b])
                world._mass = r["M"].copy()
                world._viscosity = r["B"].copy()
                world._nleffects = r["N"].copy()
            return
        if stage == "controllers":
            r = bw.inspect(tq, tdq, dt, ["Z", "gforce0"], skip_constraints=True)
            world._impedance = r["Z"].cpu().numpy()[0].copy()
            world._gforce = r["gforce0"].cpu().numpy()[0].copy()
            world._admittance = np.linalg.inv(world._impedance)
            return
        if stage == "constraints":
            self._warm = cf.copy()
            r = bw.inspect(tq, tdq, dt, ["pose", "c_sdist", "c_active", "c_force", "c_frame", "gforce"],
                           cforce=tcf)
            r = {k: v.cpu().numpy()[0] for k, v in r.items()}
            world._gforce = r["gforce"].copy()
            self._scatter_constraints(world, m, r, bodies)
            return
        if stage == "integrate":
            skip = not world._constraints_done
            if m.nc and not skip and self._warm is not None:
                tcf = torch.as_tensor(self._warm, dtype=torch.float64, device=bw.device).contiguous()
            bw.step(tq, tdq, dt, 1, cforce=None if skip else tcf, skip_constraints=skip)
            torch.cuda.synchronize(bw.device)
            q = tq.cpu().numpy()[0]
            dq = tdq.cpu().numpy()[0]
            world._gvel[:] = dq
            for b, j in enumerate(world.iterjoints()):
                qs = slice(int(m.q_off[b]), int(m.q_off[b] + m.jnq[b]))
                j.gvel = world._gvel[j.dof]
                if m.jtype[b] == JT_FREE:
                    j.gpos = q[qs].reshape(4, 4).copy()
                else:
                    j.gpos[:] = q[qs]
            self._warm = None
            return
        raise ValueError(stage)

    @staticmethod
    def _scatter_constraints(world, m, r, bodies):
        for c, con in enumerate(world._constraints):
            ct = int(m.ctype[c])
            nd = _NDOL[ct]
            if not m.c_enabled[c]:
                continue
            if ct == CT_SOFTFINGER:
                con._force = r["c_force"][c, :nd].copy()
                con._sdist = float(r["c_sdist"][c])
                con._is_active = bool(r["c_active"][c])
                for k, b in enumerate((int(m.c_body0[c]), int(m.c_body[c]))):      # constraints.py:287-288
                    Hgc = r["c_frame"][c, k]
                    con._frames[k].bpose = Hgc.copy() if b < 0 else _hinv(r["pose"][b]) @ Hgc
            elif ct == CT_JOINTLIMITS:
                con._force = r["c_force"][c, :nd].copy()
                con._pos0 = con._joint.gpos
            else:
                con._force = r["c_force"][c, :nd].copy()
                con.update(None)        # refresh _pos0 from the (device) body poses


def _body_nleffects(twist, mass):
    """core.py:1276-1288 from the device twist."""
    w = twist[0:3]
    wx = np.array([[0., -w[2], w[1]], [w[2], 0., -w[0]], [-w[1], w[0], 0.]])
    rx = np.zeros((3, 3)) if mass[3, 3] <= 1e-10 else mass[0:3, 3:6] / mass[3, 3]
    N = np.zeros((6, 6))
    N[0:3, 0:3] = wx
    N[3:6, 3:6] = wx
    N[0:3, 3:6] = rx @ wx - wx @ rx
    return N @ mass
