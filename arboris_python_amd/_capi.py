"""ctypes binding of libarbstep.so (the C ABI declared in include/arbstep.h).

The library is the product: there is no Python/NumPy fallback for the step.
``load()`` raises ``RuntimeError`` with build instructions when the shared
object is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libarbstep.so")
if os.environ.get("ARBSTEP_LIB"):                 # development: load a differently built library
    LIB_PATH = os.environ["ARBSTEP_LIB"]

ARB_ABI_VERSION = 8
ARB_OK = 0
ARB_ERR_STALLED = 5
ARB_F32, ARB_F64 = 0, 1
ARB_MAXDOL = 4
ARB_STEP_SKIP_CONSTRAINTS = 1
ARB_STEP_FUSED = 2
ARB_STEP_SPLIT_WAVE = 8
ARB_STEP_MFMA_ELIM = 16
ARB_STEP_STATIC_WORLDS = 32
ARB_STEP_WAVES2 = 64
ARB_STEP_WAVES3 = 128
ARB_STEP_ONE_WORLD = 256
ARB_STEP_GENERAL_KERNELS = 512
ARB_STEP_BODY_COLUMNS = 1024
ARB_STEP_MIXED = 2048
ARB_STEP_NO_MIXED = 4096
ARB_STEP_CLASSIC_COLUMNS = 8192
ARB_WIDE_MAX = 1024
ARB_WARN_ILLCOND = 1
ARB_WARN_ACTIVE_CONSTRAINTS = 2
ARB_WIDE_MAX_CONSTRAINTS = 256
ARB_ILLCOND_GROWTH = 2048.0

_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("nb", C.c_int32), ("ndof", C.c_int32), ("nq", C.c_int32), ("nc", C.c_int32),
        ("parent", _PI), ("jtype", _PI), ("dof_off", _PI), ("q_off", _PI),
        ("H_pr", _PD), ("H_cn", _PD), ("mass", _PD), ("visc", _PD),
        ("weighted", _PI),
        ("gravity", C.c_double * 3),
        ("up", C.c_double * 3),
        ("pd_kp", _PD), ("pd_kd", _PD), ("pd_tau0", _PD),
        ("ctype", _PI), ("c_enabled", _PI), ("c_body", _PI), ("c_body0", _PI), ("c_geom", _PI), ("c_dof", _PI),
        ("c_local", _PD), ("c_radius", _PD), ("c_radius0", _PD), ("c_half", _PD), ("c_plane", _PD),
        ("c_mu", _PD), ("c_prox", _PD), ("c_eps", _PD), ("c_min", _PD), ("c_max", _PD),
        ("c_bpose0", _PD), ("c_bpose1", _PD),
    ]


class ModelInfo(C.Structure):
    _fields_ = [("nb", C.c_int32), ("ndof", C.c_int32), ("nq", C.c_int32), ("nc", C.c_int32),
                ("nmax", C.c_int32), ("ncols", C.c_int32), ("nsets", C.c_int32),
                ("lds_bytes_f32", C.c_int32), ("lds_bytes_f64", C.c_int32), ("device", C.c_int32),
                ("forest_copies", C.c_int32), ("mixed_default", C.c_int32), ("wide", C.c_int32),
                ("rest_pivot_growth", C.c_float)]


INSPECT_FIELDS = ["pose", "twist", "jac", "djac", "M", "B", "N", "Z", "gforce0", "vel_free",
                  "c_sdist", "c_active", "c_jac", "c_force", "c_frame", "gforce", "q_next", "dq_next", "gs_stats", "energy", "stamps", "gs_trace", "c_adm", "c_vel", "pivot_growth"]


class StepPlan(C.Structure):
    _fields_ = [("waves_per_simd", C.c_int32), ("worlds_per_wavefront", C.c_int32), ("feat", C.c_int32),
                ("lds_bytes", C.c_int32), ("wave_slots", C.c_int32), ("work_queue", C.c_int32)]


class RolloutLog(C.Structure):
    _fields_ = [("q_log", C.c_void_p), ("dq_log", C.c_void_p), ("energy_log", C.c_void_p)]


class StepCost(C.Structure):
    _fields_ = [("cost_out", C.c_void_p), ("w_q", C.c_void_p), ("w_dq", C.c_void_p), ("w_tau", C.c_void_p), ("q_ref", C.c_void_p)]


class StepArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("dq", C.c_void_p), ("cforce", C.c_void_p), ("ext_gforce", C.c_void_p),
                ("pd_qdes", C.c_void_p), ("pd_dqdes", C.c_void_p), ("pd_kp", C.c_void_p), ("pd_kd", C.c_void_p),
                ("nworlds", C.c_int64), ("dt", C.c_double), ("nsteps", C.c_int32), ("flags", C.c_uint32),
                ("log", C.POINTER(RolloutLog)), ("dt_steps", C.c_void_p),
                ("ext_gforce_steps", C.c_void_p), ("pd_qdes_steps", C.c_void_p), ("pd_dqdes_steps", C.c_void_p),
                ("cost", C.POINTER(StepCost)), ("ext_impedance", C.c_void_p)]


class InspectOut(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in INSPECT_FIELDS]


# every symbol include/arbstep.h declares (tests check they are all exported)
EXPORTED = ["arb_abi_version", "arb_strerror", "arb_last_hip_error", "arb_model_create",
            "arb_model_destroy", "arb_model_get_info", "arb_model_status", "arb_model_warnings", "arb_step_plan", "arb_step", "arb_step_ex", "arb_rollout",
            "arb_inspect", "arb_inspect_ex"]
# every symbol include/arbstep_hooks.h declares: host builds of the device math (unit tests, Constraint.solve)
TEST_HOOKS = ["arb_hook_set_knob", "arb_dev_softfinger_solve", "arb_dev_eig6_pair", "arb_build_variants", "arb_host_softfinger_solve", "arb_host_softfinger_try", "arb_host_slide_root", "arb_host_real_root_cascade", "arb_host_eig6", "arb_host_block_pinv", "arb_host_joint_local",
              "arb_host_exp_twist", "arb_host_zaligned", "arb_host_narrow_phase", "arb_host_growth_bits"]

_lib = None
_variants = None
VARIANTS_PATH = os.path.join(_HERE, "libarbstep_variants.so")


def load():
    """Load libarbstep.so once; raise RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def load_variants():
    """The test library (`make -C arboris_python_amd/csrc variants`: float32 / 44-row GENERAL kernels only, sweeps that run
    the complete local solve throughout): loaded by the tests that hold the specialised kernels and the fast sweeps of
    the shipped library bit-identical (``BatchedWorlds(model, lib=_capi.load_variants())``)."""
    global _variants
    if _variants is None:
        _variants = _open(VARIANTS_PATH)
    return _variants


def _open(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built (%s missing). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C arboris_python_amd/csrc [variants]`. "
            "There is no CPU fallback for the step." % (os.path.basename(path), path))
    # The state lives in torch tensors, so the process must run on ONE HIP runtime: torch's wheel bundles its
    # own libamdhip64.so.7, and libarbstep.so needs that soname too.  Whichever is loaded first serves both;
    # if libarbstep.so came first it would pull /opt/rocm's copy, torch would then load a second runtime and
    # hipSetDevice would report "no ROCm-capable device".  Importing torch first makes its copy the only one.
    try:
        import torch  # noqa: F401
    except ImportError:                                # pragma: no cover - symbol checks still work without torch
        pass
    lib = C.CDLL(path)
    lib.arb_abi_version.restype = C.c_int
    lib.arb_strerror.restype = C.c_char_p
    lib.arb_strerror.argtypes = [C.c_int]
    lib.arb_last_hip_error.restype = C.c_char_p
    lib.arb_model_create.restype = C.c_int
    lib.arb_model_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.arb_model_destroy.restype = C.c_int
    lib.arb_model_destroy.argtypes = [C.c_void_p]
    lib.arb_model_get_info.restype = C.c_int
    lib.arb_model_get_info.argtypes = [C.c_void_p, C.POINTER(ModelInfo)]
    lib.arb_model_status.restype = C.c_int
    lib.arb_model_status.argtypes = [C.c_void_p]
    lib.arb_model_warnings.restype = C.c_int
    lib.arb_model_warnings.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.arb_hook_set_knob.restype = C.c_int
    lib.arb_hook_set_knob.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.arb_step_plan.restype = C.c_int
    lib.arb_step_plan.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_uint32, C.c_int32, C.POINTER(StepPlan)]
    lib.arb_step.restype = C.c_int
    lib.arb_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int64, C.c_double, C.c_int32, C.c_uint32, C.c_void_p]
    lib.arb_step_ex.restype = C.c_int
    lib.arb_step_ex.argtypes = [C.c_void_p, C.c_int, C.POINTER(StepArgs), C.c_void_p]
    lib.arb_rollout.restype = C.c_int
    lib.arb_rollout.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_double, C.c_int32, C.c_uint32, C.POINTER(RolloutLog), C.c_void_p]
    lib.arb_inspect.restype = C.c_int
    lib.arb_inspect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_double, C.c_uint32, C.POINTER(InspectOut), C.c_void_p]
    lib.arb_inspect_ex.restype = C.c_int
    lib.arb_inspect_ex.argtypes = [C.c_void_p, C.c_int, C.POINTER(StepArgs), C.POINTER(InspectOut), C.c_void_p]
    lib.arb_host_growth_bits.restype = C.c_int
    lib.arb_host_growth_bits.argtypes = [C.c_float, C.c_float]
    lib.arb_host_softfinger_solve.restype = C.c_int
    lib.arb_host_softfinger_solve.argtypes = [C.c_int, _PD, _PD, _PD, C.c_double, C.c_double,
                                              C.c_double, _PD, _PD]
    lib.arb_dev_softfinger_solve.restype = C.c_int
    lib.arb_dev_softfinger_solve.argtypes = [C.c_int, C.c_int, C.c_int, _PD, _PD]
    lib.arb_host_softfinger_try.restype = C.c_int
    lib.arb_host_softfinger_try.argtypes = [C.c_int, _PD, _PD, _PD, C.c_double, C.c_double, C.c_double, _PD]
    lib.arb_host_slide_root.restype = C.c_int
    lib.arb_host_slide_root.argtypes = [_PD, C.c_double, C.c_double, C.c_double, _PD]
    lib.arb_host_real_root_cascade.restype = C.c_int
    lib.arb_host_real_root_cascade.argtypes = [_PD, C.c_double, _PD]
    lib.arb_build_variants.restype = C.c_int
    lib.arb_build_variants.argtypes = []
    lib.arb_dev_eig6_pair.restype = C.c_int
    lib.arb_dev_eig6_pair.argtypes = [C.c_int, C.c_int, C.c_int, _PD, _PD]
    lib.arb_host_eig6.restype = C.c_int
    lib.arb_host_eig6.argtypes = [_PD, _PD, _PD]
    lib.arb_host_block_pinv.restype = C.c_int
    lib.arb_host_block_pinv.argtypes = [C.c_int, C.c_int, _PD, _PD]
    lib.arb_host_joint_local.restype = C.c_int
    lib.arb_host_joint_local.argtypes = [C.c_int, _PD, _PD, _PD]
    lib.arb_host_exp_twist.restype = C.c_int
    lib.arb_host_exp_twist.argtypes = [_PD, _PD]
    lib.arb_host_zaligned.restype = None
    lib.arb_host_zaligned.argtypes = [_PD, _PD]
    lib.arb_host_narrow_phase.restype = C.c_double
    lib.arb_host_narrow_phase.argtypes = [C.c_int, _PD, _PD, C.c_double, C.c_double, _PD, _PD, _PD, _PD, _PD]
    if lib.arb_abi_version() != ARB_ABI_VERSION:
        raise RuntimeError("%s: ABI version mismatch" % os.path.basename(path))
    return lib


class ArbError(RuntimeError):
    pass


def check(status):
    if status != ARB_OK:
        lib = load()
        msg = lib.arb_strerror(status).decode()
        if status == 3:
            msg += ": " + lib.arb_last_hip_error().decode()
        raise ArbError("libarbstep: %s (status %d)" % (msg, status))


def _dp(a):
    return a.ctypes.data_as(_PD)


def _ip(a):
    return a.ctypes.data_as(_PI)


def make_desc(m):
    """Fill an ``arb_model_desc`` from a ``flatten.FlatModel``.  Returns
    ``(desc, keepalive)``; ``keepalive`` owns the host arrays the desc points to."""
    keep = []

    def f64(x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        if shape is not None:
            a = a.reshape(shape)
        keep.append(a)
        return _dp(a)

    def i32(x):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.int32))
        keep.append(a)
        return _ip(a)

    d = ModelDesc()
    d.abi_version = ARB_ABI_VERSION
    d.nb, d.ndof, d.nq, d.nc = int(m.nb), int(m.ndof), int(m.nq), int(m.nc)
    d.parent, d.jtype = i32(m.parent), i32(m.jtype)
    d.dof_off, d.q_off = i32(m.dof_off), i32(m.q_off)
    d.H_pr, d.H_cn = f64(m.H_pr), f64(m.H_cn)
    d.mass, d.visc = f64(m.mass), f64(m.visc)
    d.weighted = i32(m.weighted)
    for i in range(3):
        d.gravity[i] = float(m.gravity[i])
        d.up[i] = float(m.up[i])
    if m.has_pd:
        d.pd_kp, d.pd_kd, d.pd_tau0 = f64(m.pd_kp), f64(m.pd_kd), f64(m.pd_tau0)
    if m.nc:
        d.ctype, d.c_enabled = i32(m.ctype), i32(m.c_enabled)
        d.c_body, d.c_body0, d.c_dof = i32(m.c_body), i32(m.c_body0), i32(m.c_dof)
        d.c_geom = i32(m.c_geom)
        d.c_local, d.c_radius = f64(m.c_local), f64(m.c_radius)
        d.c_radius0, d.c_half, d.c_plane = f64(m.c_radius0), f64(m.c_half), f64(m.c_plane)
        d.c_mu, d.c_prox, d.c_eps = f64(m.c_mu), f64(m.c_prox), f64(m.c_eps)
        d.c_min, d.c_max = f64(m.c_min), f64(m.c_max)
        d.c_bpose0, d.c_bpose1 = f64(m.c_bpose0), f64(m.c_bpose1)
    return d, keep
