"""BatchedWorlds: thousands of instances of one flattened world on one MI355X.

Host-side driver of the C ABI (include/arbstep.h).  State lives in PyTorch-ROCm
tensors owned by the caller (``q`` (B,nq), ``dq`` (B,ndof), optional ``cforce``
(B,nc,4)); the library only holds the immutable model.  PyTorch is plumbing
here -- device memory and streams -- the arithmetic is in csrc/arb_kernels.hip.
"""
import ctypes as C

import numpy as np

from . import _capi
from .flatten import FlatModel, flatten_world


def _torch():
    import torch
    return torch


class BatchedWorlds(object):
    """One device-resident model + helpers to step/inspect batches of states.

    ``model`` is a ``FlatModel`` or an initialised ``core.World``.
    """

    def __init__(self, model, device=0, lib=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedWorlds needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the step")
        if not isinstance(model, FlatModel):
            model = flatten_world(model)[0]
        self.model = model
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._lib = lib if lib is not None else _capi.load()      # (lib: another build of the library, tests)
        desc, keep = _capi.make_desc(model)
        handle = C.c_void_p()
        status = self._lib.arb_model_create(C.byref(desc), self.device_index, C.byref(handle))
        if status == 2:                      # ARB_ERR_UNSUPPORTED: say which limit (include/arbstep.h)
            why = []
            if model.ndof > _capi.ARB_WIDE_MAX or model.nb > _capi.ARB_WIDE_MAX:
                why.append("%d dofs / %d bodies (at most %d)" % (model.ndof, model.nb, _capi.ARB_WIDE_MAX))
            if model.nc > _capi.ARB_WIDE_MAX_CONSTRAINTS:
                why.append("%d constraints (at most %d registered, 64 active in a step)" % (model.nc, _capi.ARB_WIDE_MAX_CONSTRAINTS))
            raise _capi.ArbError("libarbstep: model not supported by the device step (status 2)%s"
                                 % (": " + "; ".join(why) if why else ": bodies not in depth-first order, or mass matrices that are "
                                    "not rigid-body inertias"))
        _capi.check(status)
        self._handle = handle
        del keep
        info = _capi.ModelInfo()
        _capi.check(self._lib.arb_model_get_info(self._handle, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in _capi.ModelInfo._fields_}

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.arb_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------
    def _dtype_code(self, t):
        torch = _torch()
        if t.dtype == torch.float32:
            return _capi.ARB_F32
        if t.dtype == torch.float64:
            return _capi.ARB_F64
        raise TypeError("state tensors must be float32 or float64")

    def _check_state(self, q, dq, cforce=None, ext=None, nsteps=None, zimp=None):
        m = self.model
        B = q.shape[0]
        ext_shape = (B, m.ndof)
        if ext is not None and ext.dim() == 3:          # a torque SEQUENCE: one row per step (arb_step_args.ext_gforce_steps)
            if nsteps is None:
                raise ValueError("ext_gforce: a torque sequence (nsteps, B, ndof) is not accepted here; pass one row (B, ndof)")
            ext_shape = (int(nsteps), B, m.ndof)
        for name, t, shape in (("q", q, (B, m.nq)), ("dq", dq, (B, m.ndof)),
                               ("cforce", cforce, (B, m.nc, _capi.ARB_MAXDOL)),
                               ("ext_gforce", ext, ext_shape), ("ext_impedance", zimp, (B, m.ndof, m.ndof))):
            if t is None:
                continue
            if tuple(t.shape) != shape:
                raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), shape))
            if not t.is_contiguous() or t.device != self.device or t.dtype != q.dtype:
                raise ValueError("%s must be a contiguous %s tensor on %s" % (name, q.dtype, self.device))
        return B

    def to_device(self, q, dq, dtype=None):
        """NumPy (B,nq)/(B,ndof) -> device tensors of ``dtype`` (default float32)."""
        torch = _torch()
        dtype = torch.float32 if dtype is None else dtype
        return (torch.as_tensor(np.ascontiguousarray(q), dtype=dtype, device=self.device).contiguous(),
                torch.as_tensor(np.ascontiguousarray(dq), dtype=dtype, device=self.device).contiguous())

    def new_cforce(self, B, dtype):
        torch = _torch()
        return torch.zeros((B, self.model.nc, _capi.ARB_MAXDOL), dtype=dtype, device=self.device)

    def _dt_steps(self, dt, nsteps, st):
        """A scalar ``dt`` (Python or NumPy number, 0-d array or tensor) -> None; a sequence / array / tensor of
        ``nsteps`` step lengths (a non-uniform timeline, core.py:1357) -> float64 device tensor for
        ``arb_step_args.dt_steps``, uploaded ON the launch stream ``st`` and kept alive on ``self`` until the
        next call replaces it (the kernel reads it asynchronously)."""
        torch = _torch()
        a = np.asarray(dt.detach().cpu() if hasattr(dt, "detach") else dt, dtype=np.float64)
        if a.ndim == 0:
            return None
        if a.ndim != 1 or a.shape[0] != int(nsteps):
            raise ValueError("dt must be a scalar or a sequence of nsteps = %d step lengths" % int(nsteps))
        if not bool((a > 0).all()):
            raise ValueError("every dt must be positive")
        host = torch.as_tensor(np.ascontiguousarray(a))
        with torch.cuda.stream(st):                 # the copy is ordered before the kernel that reads it
            t = host.to(self.device).contiguous()
        t.record_stream(st)
        self._dts_keep = (t, host)
        return t

    @staticmethod
    def _split_flag(split):
        if split in (None, False):
            return 0
        if split == "wave":
            return _capi.ARB_STEP_SPLIT_WAVE
        raise ValueError("split must be False or 'wave' (the lane-per-world sweep kernel of ABI <= 4 was removed)")

    # -- the step --------------------------------------------------------------
    @staticmethod
    def _waves_flag(waves):
        if waves is None:
            return 0
        if waves in (2, 3):
            return _capi.ARB_STEP_WAVES2 if waves == 2 else _capi.ARB_STEP_WAVES3
        raise ValueError("waves must be None (the library picks by batch size), 2 or 3")

    def set_knob(self, name, value):
        """A development / test knob of this handle (``arb_hook_set_knob``, include/arbstep_hooks.h): the library reads
        no environment variable."""
        _capi.check(self._lib.arb_hook_set_knob(self._handle, name.encode(), int(value)))

    def warnings(self):
        """Warning bits the handle's launches raised so far (``arb_model_warnings``; cleared by the call):
        ``_capi.ARB_WARN_ILLCOND`` = a float32 launch met a world whose impedance matrix float32 cannot eliminate to 1e-5
        (long serial chains): step such a model in float64."""
        w = C.c_uint32(0)
        _capi.check(self._lib.arb_model_warnings(self._handle, C.byref(w)))
        return int(w.value)

    def step(self, q, dq, dt, nsteps=1, cforce=None, ext_gforce=None, skip_constraints=False,
             stream=None, fused=False, split=False, pd_targets=None, pd_gains=None, mfma=False, static_worlds=False,
             waves=None, one_world=False, cost=None, general_kernels=False, body_columns=False, ext_impedance=None,
             mixed=None, classic_columns=False, _log=None):
        """Advance every world by ``nsteps`` steps of ``dt`` in place (asynchronous).  ``dt`` is a scalar, or one
        step length per step (``simulate`` takes ``dt = next_time - current_time`` from its timeline,
        core.py:1357): the whole non-uniform timeline then runs inside one launch.

        ``split="wave"`` runs the Gauss-Seidel sweeps in a second kernel with one wavefront per world
        (ARB_STEP_SPLIT_WAVE: bit-identical results, measured slower at every batch size; opt-in);
        the default keeps them in the step kernel (``fused`` is accepted for symmetry).
        ``mfma=True`` (float32): phase C eliminates on the matrix cores (ARB_STEP_MFMA_ELIM; slower).
        ``static_worlds=True``: one workgroup per world for the whole launch (ARB_STEP_STATIC_WORLDS) instead of the
        device-side queue of (chunk of steps, world) items that multi-step launches of large batches use by default.
        ``waves=2|3`` pins the float32 kernel build (ARB_STEP_WAVES2/3, include/arbstep.h): by default the library picks
        by batch size; the builds agree to rounding, each is bit-reproducible across launch shapes.
        ``one_world=True``: one world per wavefront even for a small model (ARB_STEP_ONE_WORLD).  By default the worlds
        of a model of at most 16 dofs share wavefronts once the batch exceeds twice the device's wave slots: the library
        steps ``B // k`` worlds of a forest of ``k = self.info["forest_copies"]`` copies of the model on the same buffers.
        ``ext_gforce`` (B,ndof): user torques, one row per world (a zero-impedance Controller), constant over the
        launch; ``ext_gforce`` (nsteps,B,ndof): a torque SEQUENCE, step t of the launch applies row t (the reference polls
        its controllers every step, core.py:811-817: an MPC horizon in one launch).
        ``pd_targets=(qdes, dqdes)`` (B,ndof) each -- or (nsteps,B,ndof) each: a target sequence --: one
        ProportionalDerivativeController target per world (controllers.py:63-158), with the model's gains, or with the
        per-world DIAGONAL gains ``pd_gains=(kp, kd)`` (B,ndof) each.
        ``cost=dict(out=(B,), w_q=(ndof,), w_dq=(ndof,), w_tau=(ndof,), q_ref=(ndof,))`` (``out`` required, the others
        optional): the running cost of the rollout, ``out[w] += sum_t sum_i w_q (q_i - q_ref_i)^2 + w_dq dq_i^2 +
        w_tau tau_t,i^2`` on the state after every step (``arb_step_cost``, include/arbstep.h), summed on chip.
        ``general_kernels=True``: the general kernels also for a model of a specialised class (ARB_STEP_GENERAL_KERNELS).
        ``body_columns=True``: constraint columns in body space wherever the model qualifies (ARB_STEP_BODY_COLUMNS: since
        round 6 the default for every qualifying model; ``classic_columns=True``, ARB_STEP_CLASSIC_COLUMNS, opts out: the
        classical columns, 1.7 % faster with four contacts, twice the float32 outliers).
        ``ext_impedance`` (B,ndof,ndof): the summed impedance ``Z_a`` of user-defined controllers, ``Z -= ext_impedance``
        (core.py:815-817), with their generalized force in ``ext_gforce``: the generic Controller plugin path (ABI 8).
        ``mixed``: None = the library's choice (float32 buffers of a model float32 cannot eliminate -- ``self.info
        ["mixed_default"]``: long serial chains -- run the build that eliminates in float64), True = ARB_STEP_MIXED for any
        model, False = ARB_STEP_NO_MIXED (plain float32; ``warnings()`` reports ARB_WARN_ILLCOND)."""
        torch = _torch()
        B = self._check_state(q, dq, cforce, ext_gforce, nsteps, ext_impedance)
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        flags = _capi.ARB_STEP_SKIP_CONSTRAINTS if skip_constraints else 0
        if fused:
            flags |= _capi.ARB_STEP_FUSED
        flags |= self._split_flag(split)
        if mfma:
            flags |= _capi.ARB_STEP_MFMA_ELIM
        if static_worlds:
            flags |= _capi.ARB_STEP_STATIC_WORLDS
        flags |= self._waves_flag(waves)
        if one_world:
            flags |= _capi.ARB_STEP_ONE_WORLD
        if general_kernels:
            flags |= _capi.ARB_STEP_GENERAL_KERNELS
        if body_columns:
            flags |= _capi.ARB_STEP_BODY_COLUMNS
        if classic_columns:
            flags |= _capi.ARB_STEP_CLASSIC_COLUMNS
        if mixed is not None:
            flags |= _capi.ARB_STEP_MIXED if mixed else _capi.ARB_STEP_NO_MIXED
        dts = self._dt_steps(dt, nsteps, st)
        ext_seq = ext_gforce is not None and ext_gforce.dim() == 3
        if (pd_targets is None and pd_gains is None and dts is None and not ext_seq and cost is None and _log is None
                and ext_impedance is None):
            _capi.check(self._lib.arb_step(
                self._handle, self._dtype_code(q), q.data_ptr(), dq.data_ptr(),
                None if cforce is None else cforce.data_ptr(),
                None if ext_gforce is None else ext_gforce.data_ptr(),
                B, float(dt), int(nsteps), flags, C.c_void_p(st.cuda_stream)))
            return
        a = _capi.StepArgs()
        a.q, a.dq = q.data_ptr(), dq.data_ptr()
        a.cforce = None if cforce is None else cforce.data_ptr()
        if ext_seq:
            a.ext_gforce_steps = ext_gforce.data_ptr()
        else:
            a.ext_gforce = None if ext_gforce is None else ext_gforce.data_ptr()
        for names, pair in ((("pd_qdes", "pd_dqdes"), pd_targets), (("pd_kp", "pd_kd"), pd_gains)):
            if pair is None:
                continue
            seq = names[0] == "pd_qdes" and pair[0].dim() == 3
            want = (int(nsteps), B, self.model.ndof) if seq else (B, self.model.ndof)
            for name, t in zip(names, pair):
                if tuple(t.shape) != want or not t.is_contiguous() or t.dtype != q.dtype or t.device != self.device:
                    raise ValueError("%s must be a contiguous %s %s tensor on %s" % (name, want, q.dtype, self.device))
                setattr(a, name + "_steps" if seq else name, t.data_ptr())
        keep = None
        if cost is not None:
            keep = _capi.StepCost()
            for key, field, shape in (("out", "cost_out", (B,)), ("w_q", "w_q", (self.model.ndof,)), ("w_dq", "w_dq", (self.model.ndof,)),
                                      ("w_tau", "w_tau", (self.model.ndof,)), ("q_ref", "q_ref", (self.model.ndof,))):
                t = cost.get(key)
                if t is None:
                    if key == "out":
                        raise ValueError("cost needs the accumulator tensor cost['out'] of shape (B,)")
                    continue
                if tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != q.dtype or t.device != self.device:
                    raise ValueError("cost[%r] must be a contiguous %s %s tensor on %s" % (key, shape, q.dtype, self.device))
                setattr(keep, field, t.data_ptr())
            a.cost = C.pointer(keep)
        a.nworlds, a.nsteps, a.flags = B, int(nsteps), flags
        if ext_impedance is not None:
            a.ext_impedance = ext_impedance.data_ptr()
        if _log is not None:
            a.log = C.pointer(_log)
        if dts is None:
            a.dt = float(dt)
        else:
            a.dt, a.dt_steps = 0., dts.data_ptr()
        _capi.check(self._lib.arb_step_ex(self._handle, self._dtype_code(q), C.byref(a), C.c_void_p(st.cuda_stream)))

    def rollout(self, q, dq, dt, nsteps, cforce=None, ext_gforce=None, log_state=True, log_energy=True,
                skip_constraints=False, stream=None, fused=False, split=False, waves=None, one_world=False, **step_kw):
        """Advance ``nsteps`` steps in ONE launch and return the per-step logs an Observer
        would have recorded (state and energies at the beginning of every step):
        ``{"q": (nsteps,B,nq), "dq": (nsteps,B,ndof), "energy": (nsteps,B,2)}``.  Other keywords (``pd_targets``,
        ``pd_gains``, ``cost``, a torque sequence as ``ext_gforce`` ...) as for ``step``."""
        torch = _torch()
        B = self._check_state(q, dq, cforce, ext_gforce, nsteps)
        m = self.model
        out = {}
        log = _capi.RolloutLog()
        if log_state:
            out["q"] = torch.empty((nsteps, B, m.nq), dtype=q.dtype, device=self.device)
            out["dq"] = torch.empty((nsteps, B, m.ndof), dtype=q.dtype, device=self.device)
            log.q_log, log.dq_log = out["q"].data_ptr(), out["dq"].data_ptr()
        if log_energy:
            out["energy"] = torch.empty((nsteps, B, 2), dtype=q.dtype, device=self.device)
            log.energy_log = out["energy"].data_ptr()
        scalar_dt = np.ndim(dt.detach().cpu() if hasattr(dt, "detach") else dt) == 0
        if scalar_dt and not step_kw and not (ext_gforce is not None and ext_gforce.dim() == 3):
            # (the plain entry point: arb_rollout)
            st = torch.cuda.current_stream(self.device) if stream is None else stream
            flags = _capi.ARB_STEP_SKIP_CONSTRAINTS if skip_constraints else 0
            if fused:
                flags |= _capi.ARB_STEP_FUSED
            flags |= self._split_flag(split) | self._waves_flag(waves) | (_capi.ARB_STEP_ONE_WORLD if one_world else 0)
            _capi.check(self._lib.arb_rollout(
                self._handle, self._dtype_code(q), q.data_ptr(), dq.data_ptr(),
                None if cforce is None else cforce.data_ptr(),
                None if ext_gforce is None else ext_gforce.data_ptr(),
                B, float(dt), int(nsteps), flags, C.byref(log), C.c_void_p(st.cuda_stream)))
            return out
        self.step(q, dq, dt, nsteps, cforce=cforce, ext_gforce=ext_gforce, skip_constraints=skip_constraints, stream=stream,
                  fused=fused, split=split, waves=waves, one_world=one_world, _log=log, **step_kw)
        return out

    def plan(self, nworlds, nsteps=1, dtype=None, ext_gforce=False, other_inputs=False, waves=None, split=False,
             static_worlds=False, one_world=False, world_logs=False, general_kernels=False, body_columns=False, cost=False,
             mixed=None, classic_columns=False):
        """Which kernel build and launch shape ``step`` would use (``arb_step_plan``): a dict with ``waves_per_simd``,
        ``worlds_per_wavefront`` (the copies of a small model's forest), ``feat``, ``lds_bytes``, ``wave_slots``, ``work_queue``.
        ``world_logs``: the launch is a rollout that logs per-world energies (or states of a batch that is not a multiple
        of the forest's copies), which a small model runs one world per wavefront.  ``cost``: the launch carries a running
        cost (per world, like the energies: a small model runs one world per wavefront; it travels with the user torques)."""
        torch = _torch()
        world_logs = world_logs or bool(cost)
        ext_gforce = ext_gforce or bool(cost)
        code = _capi.ARB_F64 if dtype == torch.float64 else _capi.ARB_F32
        flags = self._waves_flag(waves) | self._split_flag(split) | (_capi.ARB_STEP_STATIC_WORLDS if static_worlds else 0)
        flags |= _capi.ARB_STEP_ONE_WORLD if one_world else 0
        flags |= _capi.ARB_STEP_GENERAL_KERNELS if general_kernels else 0
        flags |= _capi.ARB_STEP_BODY_COLUMNS if body_columns else 0
        flags |= _capi.ARB_STEP_CLASSIC_COLUMNS if classic_columns else 0
        if mixed is not None:
            flags |= _capi.ARB_STEP_MIXED if mixed else _capi.ARB_STEP_NO_MIXED
        p = _capi.StepPlan()
        _capi.check(self._lib.arb_step_plan(self._handle, code, int(nworlds), int(nsteps), flags,
                                            (3 if other_inputs else (1 if ext_gforce else 0)) | (4 if world_logs else 0), C.byref(p)))
        return {k: getattr(p, k) for k, _ in _capi.StepPlan._fields_}

    def status(self):
        """Health of this handle's launches so far (``arb_model_status``): raises ``ArbError`` (ARB_ERR_STALLED) when a
        launch gave up waiting inside its device-side work queue -- its results are invalid.  Reads host memory only;
        synchronise the stream first to learn about launches that are still queued.

        The stall word is STICKY: once a launch has stalled, every ``step`` / ``rollout`` / ``inspect`` call on this
        handle raises ARB_ERR_STALLED without launching until THIS method has been called -- it is the acknowledgement.
        The protocol after catching the error: synchronise, call ``status()`` (it raises once more and clears the word),
        RELOAD the states (the stalled launch left them half-advanced), then step again.  Retrying ``step`` without
        calling ``status()`` raises forever."""
        _capi.check(self._lib.arb_model_status(self._handle))

    def acknowledge_stall(self):
        """``status()`` without the exception: clears a raised stall word and returns True when one had been recorded
        (the states of the stalled launch are invalid: reload them before stepping on)."""
        return self._lib.arb_model_status(self._handle) == _capi.ARB_ERR_STALLED

    def inspect(self, q, dq, dt, want, cforce=None, ext_gforce=None, skip_constraints=False, general_kernels=False,
                body_columns=False, ext_impedance=None, pd_targets=None, pd_gains=None, classic_columns=False):
        """Evaluate one step without touching ``q``/``dq``; returns a dict of the
        requested intermediate results (names of ``arb_inspect_out``).  ``general_kernels`` / ``body_columns``: the
        arithmetic of ``step`` with the same flag (the inspect kernel forms the constraint-space system the same way).
        ``ext_impedance``, ``pd_targets``, ``pd_gains`` as for ``step`` (``arb_inspect_ex``, ABI 8)."""
        torch = _torch()
        m = self.model
        B = self._check_state(q, dq, cforce, ext_gforce, None, ext_impedance)
        n, nb, nc, nq = m.ndof, m.nb, m.nc, m.nq
        shapes = dict(pose=(B, nb, 4, 4), twist=(B, nb, 6), jac=(B, nb, 6, n), djac=(B, nb, 6, n),
                      M=(B, n, n), B=(B, n, n), N=(B, n, n), Z=(B, n, n), gforce0=(B, n),
                      vel_free=(B, n), c_sdist=(B, nc), c_active=(B, nc), c_jac=(B, nc, 4, n),
                      c_force=(B, nc, 4), c_frame=(B, nc, 2, 4, 4), gforce=(B, n),
                      q_next=(B, nq), dq_next=(B, n), gs_stats=(B, 5), stamps=(B, 8), energy=(B, 2),
                      gs_trace=(B, 20, nc), c_adm=(B, 4 * nc, 4 * nc), c_vel=(B, 4 * nc), pivot_growth=(B,))
        want = list(want)
        if "gforce" in want and nc and "c_jac" not in want:
            want.append("c_jac")
        out = _capi.InspectOut()
        res = {}
        for name in want:
            dt_ = torch.int32 if name in ("c_active", "gs_stats", "gs_trace") else (torch.int64 if name == "stamps" else q.dtype)
            t = torch.zeros(shapes[name], dtype=dt_, device=self.device)
            if name == "gs_trace":
                t.fill_(-1)                       # solves that are not executed leave their entry alone
            res[name] = t
            setattr(out, name, t.data_ptr())
        st = torch.cuda.current_stream(self.device)
        flags = _capi.ARB_STEP_SKIP_CONSTRAINTS if skip_constraints else 0
        flags |= _capi.ARB_STEP_GENERAL_KERNELS if general_kernels else 0
        flags |= _capi.ARB_STEP_BODY_COLUMNS if body_columns else 0
        flags |= _capi.ARB_STEP_CLASSIC_COLUMNS if classic_columns else 0
        if ext_impedance is None and pd_targets is None and pd_gains is None:
            _capi.check(self._lib.arb_inspect(
                self._handle, self._dtype_code(q), q.data_ptr(), dq.data_ptr(),
                None if cforce is None else cforce.data_ptr(),
                None if ext_gforce is None else ext_gforce.data_ptr(),
                B, float(dt), flags, C.byref(out), C.c_void_p(st.cuda_stream)))
            return res
        a = _capi.StepArgs()
        a.q, a.dq = q.data_ptr(), dq.data_ptr()
        a.cforce = None if cforce is None else cforce.data_ptr()
        a.ext_gforce = None if ext_gforce is None else ext_gforce.data_ptr()
        a.ext_impedance = None if ext_impedance is None else ext_impedance.data_ptr()
        for names, pair in ((("pd_qdes", "pd_dqdes"), pd_targets), (("pd_kp", "pd_kd"), pd_gains)):
            if pair is None:
                continue
            for name, t in zip(names, pair):
                if tuple(t.shape) != (B, n) or not t.is_contiguous() or t.dtype != q.dtype or t.device != self.device:
                    raise ValueError("%s must be a contiguous %s %s tensor on %s" % (name, (B, n), q.dtype, self.device))
                setattr(a, name, t.data_ptr())
        a.nworlds, a.nsteps, a.flags, a.dt = B, 1, flags, float(dt)
        _capi.check(self._lib.arb_inspect_ex(self._handle, self._dtype_code(q), C.byref(a), C.byref(out), C.c_void_p(st.cuda_stream)))
        return res
