"""Device evaluation of the four World step methods for ONE world.

``core.World.update_dynamic / update_controllers / update_constraints /
integrate`` call ``SingleWorldEngine.run``: the world is flattened, evaluated on
the GPU in float64 as a batch of one through ``arb_inspect_ex`` / ``arb_step_ex`` and
the results are scattered back onto the objects exactly where the reference leaves
them (core.py:1272-1288 body attributes, :722-734 world matrices, :812-818
impedance/admittance, :910-937 constraint state, :974-980 joint state).

Two attributes are derived on the host from device results because the device
never forms them: ``Body.nleffects`` (from the device twist, core.py:1276-1288)
and ``World._admittance`` (inverse of the device impedance; the kernels solve
with Z instead of inverting it).

User-defined ``Controller`` subclasses (the reference's plugin point, core.py:327-339)
cannot be lowered to the kernels: ``update_controllers`` calls their ``update(dt)`` on
the host, exactly where the reference does (core.py:814-817), and hands the sums of
their ``(gforce, impedance)`` to the device as ``ext_gforce`` / ``ext_impedance``
(ABI 8) for the rest of the step.

Round 6: the flattened model is cached on a signature of everything ``flatten_world``
reads except the state (tree, frames, masses, plugin parameters), so a ``simulate`` step
flattens once instead of four times, and ``update_controllers`` fetches what
``update_constraints`` will ask for in the same launch (three launches per step
instead of four).
"""
import numpy as np

from . import _capi
from .batch import BatchedWorlds
from .flatten import (flatten_world, JT_FREE, CT_SOFTFINGER, CT_JOINTLIMITS,
                      CT_BALLSOCKET)

_NDOL = {CT_SOFTFINGER: 4, CT_JOINTLIMITS: 1, CT_BALLSOCKET: 3}
_CONSTRAINT_OUTPUTS = ["pose", "c_sdist", "c_active", "c_force", "c_frame", "gforce"]


def _hinv(H):
    out = np.eye(4)
    out[0:3, 0:3] = H[0:3, 0:3].T
    out[0:3, 3] = -H[0:3, 0:3].T @ H[0:3, 3]
    return out


def _bytes(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64)).tobytes()


def model_signature(world):
    """Everything ``flatten_world`` reads from ``world`` EXCEPT the state (joint positions / velocities, constraint
    forces): identity and type of every joint in dof order, the frames' offsets, the bodies' mass and viscosity
    matrices, and the parameters of every plugin.  Two calls with equal signatures flatten to the same model."""
    sig = [tuple(world.up)]
    for j in world.iterjoints():
        body = j._frame1.body
        sig.append((id(j), type(j).__name__, _bytes(j._frame0.bpose), _bytes(j._frame1.bpose),
                    _bytes(body.mass), _bytes(body.viscosity)))
    for a in world._controllers:
        entry = [id(a), type(a).__name__]
        for name in ("gravity", "kp", "kd", "gpos_des", "gvel_des"):
            if hasattr(a, name):
                entry.append(_bytes(getattr(a, name)))
        if hasattr(a, "joints"):
            entry.append(tuple(id(j) for j in a.joints))
        sig.append(tuple(entry))
    for c in world._constraints:
        entry = [id(c), type(c).__name__, bool(c.is_enabled())]
        for name in ("_mu", "_proximity", "_eps", "_min", "_max"):
            if hasattr(c, name):
                entry.append(_bytes(getattr(c, name)))
        for s in getattr(c, "_shapes", ()):
            entry.append((type(s).__name__, id(s.frame.body), _bytes(s.frame.bpose)) +
                         tuple(_bytes(getattr(s, k)) for k in ("radius", "half_extents", "coeffs") if hasattr(s, k)))
        if type(c).__name__ == "BallAndSocketConstraint":
            entry.extend((id(f.body), _bytes(f.bpose)) for f in c._frames)
        sig.append(tuple(entry))
    return tuple(sig)


class SingleWorldEngine(object):
    def __init__(self, world, device=0):
        self._device = device
        self._bw = None
        self._sig = None
        self._host_controllers = []
        self._warm = None
        self._ext = None                # (gforce (1,n) tensor, impedance (1,n,n) tensor) of the host controllers, this step
        self._ahead = None              # (state key, dt, results): what update_controllers fetched for update_constraints
        self.flatten_count = 0          # (diagnostic; tests/test_host_api.py)

    # -- model and state -------------------------------------------------------
    def _prepare(self, world):
        import torch
        sig = model_signature(world)
        if self._bw is None or sig != self._sig:
            host = []
            m, q, dq = flatten_world(world, host_controllers=host)
            self.flatten_count += 1
            if self._bw is not None:
                self._bw.close()
            self._bw = BatchedWorlds(m, self._device)
            self._sig, self._host_controllers = sig, host
        else:
            m = self._bw.model
            q = np.concatenate([np.asarray(j.gpos, np.float64).ravel() for j in world.iterjoints()])
            dq = np.concatenate([np.asarray(j.gvel, np.float64).ravel() for j in world.iterjoints()])
        bw = self._bw
        tq, tdq = bw.to_device(q[None], dq[None], torch.float64)
        cf = np.zeros((1, m.nc, _capi.ARB_MAXDOL))
        for c, con in enumerate(world._constraints):
            f = np.asarray(con._force, float).ravel()
            cf[0, c, :len(f)] = f
        return m, bw, tq, tdq, cf, (q.tobytes(), dq.tobytes(), cf.tobytes())

    def _poll_host_controllers(self, world, dt, bw):
        """core.py:814-817 for the controllers the kernels do not know: ``gforce += gforce_a; impedance -= Z_a``."""
        import torch
        if not self._host_controllers:
            self._ext = None
            return
        n = world.ndof
        g, z = np.zeros(n), np.zeros((n, n))
        for a in self._host_controllers:
            (ga, za) = a.update(dt)
            g += np.asarray(ga, float).reshape(n)
            z += np.asarray(za, float).reshape(n, n)
        self._ext = (torch.as_tensor(g[None], dtype=torch.float64, device=bw.device).contiguous(),
                     torch.as_tensor(z[None], dtype=torch.float64, device=bw.device).contiguous())

    def run(self, world, stage, dt):
        import torch
        m, bw, tq, tdq, cf, key = self._prepare(world)
        bodies = list(world.ground.iter_descendant_bodies())
        n = m.ndof
        tcf = torch.as_tensor(cf, dtype=torch.float64, device=bw.device).contiguous() if m.nc else None
        ext_g, ext_z = self._ext if self._ext is not None else (None, None)
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
                self._ext = None            # (the controllers have not been polled for this state yet)
            return
        if stage == "controllers":
            self._poll_host_controllers(world, dt, bw)
            ext_g, ext_z = self._ext if self._ext is not None else (None, None)
            # one launch serves this stage and the next: Z and the controllers' gforce now, the constraint outputs
            # kept for update_constraints at the same state and dt
            want = ["Z", "gforce0"] + (_CONSTRAINT_OUTPUTS if m.nc else [])
            r = bw.inspect(tq, tdq, dt, want, cforce=tcf, ext_gforce=ext_g, ext_impedance=ext_z)
            r = {k: v.cpu().numpy()[0] for k, v in r.items()}
            world._impedance = r["Z"].copy()
            world._gforce = r["gforce0"].copy()
            world._admittance = np.linalg.inv(world._impedance)
            self._ahead = (key, float(dt), r) if m.nc else None
            return
        if stage == "constraints":
            self._warm = cf.copy()
            if self._ahead is not None and self._ahead[0] == key and self._ahead[1] == float(dt):
                r = self._ahead[2]
            else:
                r = bw.inspect(tq, tdq, dt, _CONSTRAINT_OUTPUTS, cforce=tcf, ext_gforce=ext_g, ext_impedance=ext_z)
                r = {k: v.cpu().numpy()[0] for k, v in r.items()}
            self._ahead = None
            world._gforce = r["gforce"].copy()
            self._scatter_constraints(world, m, r, bodies)
            return
        if stage == "integrate":
            skip = not world._constraints_done
            if m.nc and not skip and self._warm is not None:
                tcf = torch.as_tensor(self._warm, dtype=torch.float64, device=bw.device).contiguous()
            bw.step(tq, tdq, dt, 1, cforce=None if skip else tcf, skip_constraints=skip, ext_gforce=ext_g,
                    ext_impedance=ext_z)
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
            self._ext = None
            self._ahead = None
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
