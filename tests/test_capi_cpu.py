"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/arbstep.h declares, validates its arguments, and its device math
(compiled for the host through the self-test hooks) reproduces the reference's
captured SoftFingerContact.solve tuples.  No GPU compute is launched here.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, load_model
from arboris_python_amd import _capi
import arb_oracle as O

needs_lib = pytest.mark.skipif(not os.path.exists(_capi.LIB_PATH),
                               reason="libarbstep.so not built (run __graft_entry__.build())")


@needs_lib
def test_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "arbstep.h")).read()
    declared = set(re.findall(r"\b(arb_[a-z_0-9]+)\s*\(", header))
    declared -= {"arb_model_desc", "arb_inspect_out", "arb_model_info"}
    assert declared == set(_capi.EXPORTED), declared ^ set(_capi.EXPORTED)
    # the unit-test / single-constraint hooks have a header of their own since round 3: everything the product or the
    # tests bind must be declared in one of the two
    hooks_h = open(os.path.join(ROOT, "include", "arbstep_hooks.h")).read()
    hooks = set(re.findall(r"\b(arb_[a-z_0-9]+)\s*\(", hooks_h))
    assert hooks == set(_capi.TEST_HOOKS), hooks ^ set(_capi.TEST_HOOKS)
    lib = C.CDLL(_capi.LIB_PATH)
    for name in _capi.EXPORTED + _capi.TEST_HOOKS:
        assert hasattr(lib, name), name
    # and nothing else is exported under the library's prefix
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("arb_") and " T " in l}
    assert exported == set(_capi.EXPORTED) | set(_capi.TEST_HOOKS), exported ^ (set(_capi.EXPORTED) | set(_capi.TEST_HOOKS))
    assert _capi.load().arb_abi_version() == _capi.ARB_ABI_VERSION
    assert _capi.load().arb_strerror(0) == b"ok"


@needs_lib
def test_argument_validation_without_gpu():
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.arb_model_create(None, 0, C.byref(h)) == 1          # ARB_ERR_INVALID
    m, _, _ = load_model("human36_c4")
    desc, keep = _capi.make_desc(m)
    desc.abi_version = 99
    assert lib.arb_model_create(C.byref(desc), 0, C.byref(h)) == 1
    assert lib.arb_step(None, 0, None, None, None, None, 1, 1e-3, 1, 0, None) == 1
    assert lib.arb_step(None, 0, None, None, None, None, 0, 1e-3, 1, 0, None) == 1     # still needs a model
    assert lib.arb_step_ex(None, 0, None, None) == 1
    a = _capi.StepArgs()
    a.nworlds, a.dt, a.nsteps = 1, 1e-3, 1
    assert lib.arb_step_ex(None, 0, C.byref(a), None) == 1
    assert lib.arb_model_destroy(None) == 1


@needs_lib
def test_body_numbering_must_be_dfs_preorder():
    """World.init numbers bodies depth first (core.py:611-615); the kernels rely on contiguous subtrees
    (prefix-scan subtree sums) and arb_model_create refuses anything else before touching the device."""
    lib = _capi.load()
    h = C.c_void_p()
    m, _, _ = load_model("human36_g")
    desc, keep = _capi.make_desc(m)
    parent = np.array(m.parent, dtype=np.int32).copy()
    assert parent[1] == 0 and parent[-1] not in (0, 1)
    parent[-1] = 1                      # the last body hangs below body 1, whose subtree ended long before
    keep.append(parent)
    desc.parent = parent.ctypes.data_as(C.POINTER(C.c_int32))
    assert lib.arb_model_create(C.byref(desc), 0, C.byref(h)) == 2          # ARB_ERR_UNSUPPORTED
    assert not h.value


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libarbstep.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load()


def test_world_step_without_gpu_raises():
    """The object API never falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from arboris_python_amd.core import simplearm
    w = simplearm()
    with pytest.raises(RuntimeError):
        w.update_dynamic()


@needs_lib
@pytest.mark.parametrize("route", [0, 0x100])
@pytest.mark.parametrize("branch,code", [("release", 0), ("static", 1), ("sliding", 2)])
def test_device_softfinger_solve_on_host(branch, code, route):
    """arb_math.h::softfinger_solve (float64, host build) vs the reference's
    captured solve() tuples.  route 0 = sextic + Laguerre fast path for the sliding
    shift, 0x100 = generic 6x6 QR eigenvalues (the fallback)."""
    lib = _capi.load()
    g = load_golden("g3_contacts.npz")
    n = len(g["solve_%s_dt" % branch])
    worst = 0.
    for i in range(n):
        vel = np.ascontiguousarray(g["solve_%s_vel" % branch][i])
        adm = np.ascontiguousarray(g["solve_%s_adm" % branch][i])
        f = np.ascontiguousarray(g["solve_%s_force" % branch][i].copy())
        df = np.zeros(4)
        eps = np.ones(3)
        br = lib.arb_host_softfinger_solve(_capi.ARB_F64 | route, _capi._dp(vel), _capi._dp(adm), _capi._dp(f),
                                           float(g["solve_%s_sdist" % branch][i]),
                                           float(g["solve_%s_dt" % branch][i]),
                                           float(g["solve_%s_mu" % branch][i]), _capi._dp(eps), _capi._dp(df))
        assert br == code
        ref = g["solve_%s_dforce" % branch][i]
        worst = max(worst, np.abs(df - ref).max() / max(1., np.abs(ref).max()))
    assert worst < 1e-8, worst


@needs_lib
def test_softfinger_solve_on_host_over_an_oracle_episode():
    """The local solve the kernels run (arb_math.h, host build) over every SoftFingerContact.solve call of a
    falling episode stepped by the oracle (8 worlds x 40 steps of config 3, ~1e4 tuples, all three branches):
    float64 to 1e-8; decisions may only differ where the oracle's inequality is at rounding distance."""
    import arb_oracle as O
    from conftest import load_model
    from arboris_python_amd import synth
    lib = _capi.load()
    m, _, _ = load_model("human36_c4")
    q, dq = synth.standing_states(m, 8, seed=1000, drop=0.03, vel=0.1)
    tr, cf = [], None
    for _ in range(40):
        q, dq, cf = O.step(m, q, dq, 5e-3, cf, trace=tr)
    cnt = np.bincount([t["branch"] for t in tr], minlength=3)
    assert len(tr) > 5000 and (cnt > 100).all(), cnt
    eps = np.ones(3)
    worst = 0.
    for t in tr:
        vel, adm, f = np.ascontiguousarray(t["vel"]), np.ascontiguousarray(t["adm"]), t["force"].copy()
        df = np.zeros(4)
        br = lib.arb_host_softfinger_solve(_capi.ARB_F64, _capi._dp(vel), _capi._dp(adm), _capi._dp(f), t["sdist"], t["dt"],
                                           t["mu"], _capi._dp(eps), _capi._dp(df))
        assert br == t["branch"], (br, t["branch"], t["sweep"], t["world"])
        worst = max(worst, np.abs(df - t["dforce"]).max() / max(1., np.abs(t["dforce"]).max()))
    assert worst < 1e-8, worst


@needs_lib
@pytest.mark.parametrize("dtype", [_capi.ARB_F64, _capi.ARB_F32])
def test_sliding_shift_fast_path_is_taken(dtype):
    """The register-only sextic/Laguerre route must handle every captured sliding
    case (code 2); code 3 would mean the slow generic eig6 fallback."""
    lib = _capi.load()
    g = load_golden("g3_contacts.npz")
    eps = np.ones(3)
    for i in range(len(g["solve_sliding_dt"])):
        br = lib.arb_host_softfinger_try(
            dtype, _capi._dp(np.ascontiguousarray(g["solve_sliding_vel"][i])),
            _capi._dp(np.ascontiguousarray(g["solve_sliding_adm"][i])),
            _capi._dp(np.ascontiguousarray(g["solve_sliding_force"][i])),
            float(g["solve_sliding_sdist"][i]), float(g["solve_sliding_dt"][i]),
            float(g["solve_sliding_mu"][i]), _capi._dp(eps))
        assert br == 2


@needs_lib
def test_sliding_root_formula_and_warm_start():
    """The structured sextic (per-step precompute + per-sweep scalars) has the eigenvalues
    of the reference's 6x6 matrix, and a warm start left or right of the root, certified or
    not, never changes the answer."""
    lib = _capi.load()
    g = load_golden("g3_contacts.npz")
    for i in range(0, len(g["solve_sliding_dt"]), 3):
        vel, adm, f = g["solve_sliding_vel"][i], g["solve_sliding_adm"][i], g["solve_sliding_force"][i]
        sd, dt, mu = g["solve_sliding_sdist"][i], g["solve_sliding_dt"][i], g["solve_sliding_mu"][i]
        alpha = vel - adm @ f
        alpha[3] += sd / dt
        Yc, yn = adm[0:3, 3], adm[3, 3]
        beta = alpha[0:3] - alpha[3] / yn * Yc
        a = mu / yn * alpha[3]
        b = mu / yn * Yc
        c0, c1 = Yc @ Yc / yn, 2 / a * (beta @ b)
        c2, c3m1 = (beta @ beta) / a ** 2, b @ b - 1
        B = np.zeros((6, 6))
        B[3:, 3:] = adm[0:3, 0:3] - c0
        B[:3, :3] = adm[0:3, 0:3] - c0 + c1
        B[:3, 3:] = -np.eye(3) * c2
        B[3:, :3] = np.eye(3) * c3m1
        ev = np.linalg.eigvals(B)
        ref = np.min(ev.real[np.abs(ev.imag) < 1e-12])
        Y = np.ascontiguousarray(adm)
        for warm in (float("nan"), ref, ref * 1.5, ref * 0.5, 0.3, -50.0):
            root = np.zeros(1)
            assert lib.arb_host_slide_root(_capi._dp(Y), float(c1), float(c2 * c3m1), float(warm), _capi._dp(root)) == 1
            assert abs(root[0] - ref) < 1e-9 * max(1e-3, abs(ref)), (i, warm, root[0], ref)


@needs_lib
def test_real_root_cascade_against_numpy_roots():
    """The derivative cascade (arb_math.h: slide_real_root_cascade; what decides the sliding shift when the warm-started
    iteration declines) finds the leftmost real root <= 0 of a sextic, or reports that there is none: against
    numpy.roots on sextics built from prescribed roots -- all real, real roots behind a complex pair (the case the iteration
    declines), complex only, clusters over ten decades, a root at zero, roots right of zero only."""
    lib = _capi.load()
    rng = np.random.default_rng(5)
    checked = {"real": 0, "none": 0}
    for trial in range(4000):
        kind = trial % 6
        scale = 10. ** rng.uniform(-4, 4)
        if kind == 0:                                   # six real roots, any sign
            roots = list(rng.normal(0., 1., 6) * scale)
        elif kind == 1:                                 # a complex pair LEFT of the real roots
            re = -abs(rng.normal(3., 1.)) * scale
            roots = [complex(re, abs(rng.normal()) * scale + 1e-3 * scale), None] + list(-np.abs(rng.normal(0., 1., 4)) * scale * 0.5)
        elif kind == 2:                                 # complex pairs only
            roots = []
            for _ in range(3):
                roots += [complex(rng.normal() * scale, abs(rng.normal()) * scale + 1e-2 * scale), None]
        elif kind == 3:                                 # two complex pairs and two real roots
            roots = [complex(rng.normal() * scale, abs(rng.normal()) * scale + 1e-2 * scale), None,
                     complex(-abs(rng.normal()) * scale, abs(rng.normal()) * scale + 1e-2 * scale), None] + list(rng.normal(0., 1., 2) * scale)
        elif kind == 4:                                 # roots spread over many decades, all negative
            roots = list(-10. ** rng.uniform(-6, 3, 6))
        else:                                           # positive real roots only (none admissible) or one at zero
            roots = list(np.abs(rng.normal(0., 1., 6)) * scale + 1e-3 * scale)
            if trial % 12 == 5:
                roots[0] = 0.
        full = []
        for r in roots:
            if r is None:
                full.append(np.conj(full[-1]))
            else:
                full.append(r)
        pc = np.real(np.poly(full))[::-1].copy()        # constant term first, monic
        assert pc.shape == (7,) and pc[6] == 1.
        lo = -1.0001 * (1. + np.abs(pc[:6]).max())      # Cauchy bound: left of every root
        real = sorted(float(np.real(r)) for r in full if np.imag(r) == 0. and np.real(r) <= 0.)
        root = np.zeros(1)
        rc = lib.arb_host_real_root_cascade(_capi._dp(np.ascontiguousarray(pc)), float(lo), _capi._dp(root))
        # (the prescribed roots are the reference; multiplying them out conditions the leftmost real one at about
        # 1e-16 * |largest root|^6 / |p'|, so compare through the polynomial's own residual)
        if real:
            assert rc == 1, (trial, kind, rc, real[0])
            want = real[0]
            dp = abs(np.polyval(np.polyder(pc[::-1]), want))
            bound = 1e-12 * max(abs(want), 1e-300) + 1e-13 * np.sum(np.abs(pc) * np.abs(want) ** np.arange(7)) / max(dp, 1e-300)
            assert abs(root[0] - want) <= bound, (trial, kind, root[0], want, bound)
            checked["real"] += 1
        else:
            assert rc == 0, (trial, kind, rc, root[0])
            checked["none"] += 1
    assert checked["real"] > 1500 and checked["none"] > 600, checked
    assert lib.arb_host_real_root_cascade(_capi._dp(np.array([1., np.nan, 0., 0., 0., 0., 1.])), -5., _capi._dp(np.zeros(1))) == -1


@needs_lib
def test_device_softfinger_solve_float32_on_host():
    lib = _capi.load()
    g = load_golden("g3_contacts.npz")
    worst = 0.
    for branch, code in (("static", 1), ("sliding", 2)):
        for i in range(len(g["solve_%s_dt" % branch])):
            vel = np.ascontiguousarray(g["solve_%s_vel" % branch][i])
            adm = np.ascontiguousarray(g["solve_%s_adm" % branch][i])
            f = np.ascontiguousarray(g["solve_%s_force" % branch][i].copy())
            df = np.zeros(4)
            eps = np.ones(3)
            br = lib.arb_host_softfinger_solve(_capi.ARB_F32, _capi._dp(vel), _capi._dp(adm), _capi._dp(f),
                                               float(g["solve_%s_sdist" % branch][i]),
                                               float(g["solve_%s_dt" % branch][i]),
                                               float(g["solve_%s_mu" % branch][i]), _capi._dp(eps), _capi._dp(df))
            ref = g["solve_%s_dforce" % branch][i]
            if br == code:          # a float32 borderline case may legitimately flip branch
                worst = max(worst, np.abs(df - ref).max() / max(1., np.abs(ref).max()))
    assert worst < 5e-3, worst


@needs_lib
def test_device_eig6_on_host():
    lib = _capi.load()
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(6, 6))
        wr, wi = np.zeros(6), np.zeros(6)
        assert lib.arb_host_eig6(_capi._dp(np.ascontiguousarray(A)), _capi._dp(wr), _capi._dp(wi)) == 6
        ref = np.linalg.eigvals(A)
        got = wr + 1j * wi
        for lam in ref:
            assert np.min(np.abs(got - lam)) < 1e-9 * max(1, abs(lam))


@needs_lib
def test_degenerate_sliding_inputs_terminate():
    """alpha[3] == 0 makes the reference divide by zero (numpy.linalg.eigvals then raises
    LinAlgError); the device math must neither hang nor crash: non-finite matrices have no
    eigenvalues and the sliding branch falls back to s = -1e10."""
    lib = _capi.load()
    for bad in (np.inf, -np.inf, np.nan):
        A = np.eye(6); A[2, 3] = bad
        wr, wi = np.zeros(6), np.zeros(6)
        assert lib.arb_host_eig6(_capi._dp(np.ascontiguousarray(A)), _capi._dp(wr), _capi._dp(wi)) == 0
    adm = np.diag([5., 4e-3, 4e-3, 3e-3]); adm[0:3, 3] = adm[3, 0:3] = 1e-4
    for dtype in (_capi.ARB_F64, _capi.ARB_F32, _capi.ARB_F64 | 0x100):
        f = np.zeros(4); df = np.zeros(4)
        # vel[3] + sdist/dt == 0 with zero force: alpha[3] == 0 exactly; tangential velocity forces sliding
        vel = np.array([0., 1., 0., 1.0])
        br = lib.arb_host_softfinger_solve(dtype, _capi._dp(vel), _capi._dp(np.ascontiguousarray(adm)), _capi._dp(f),
                                           -5e-3, 5e-3, 0.6, _capi._dp(np.ones(3)), _capi._dp(df))
        assert br in (1, 2)


@needs_lib
@pytest.mark.parametrize("tid", range(9))
def test_device_joint_local_on_host(tid):
    lib = _capi.load()
    g = load_golden("g0_primitives.npz")
    for q, dq, pose, jac, djac, tw in zip(g["joint%d_q" % tid], g["joint%d_dq" % tid],
                                          g["joint%d_pose" % tid], g["joint%d_jac" % tid],
                                          g["joint%d_djac" % tid], g["joint%d_twist" % tid]):
        out = np.zeros(36)
        lib.arb_host_joint_local(tid, _capi._dp(np.ascontiguousarray(q)),
                                 _capi._dp(np.ascontiguousarray(dq)), _capi._dp(out))
        assert np.abs(out[0:9].reshape(3, 3) - pose[0:3, 0:3]).max() < 1e-14
        assert np.abs(out[9:12] - pose[0:3, 3]).max() < 1e-14
        assert np.abs(out[30:36] - tw).max() < 1e-13
        if tid not in (0, 8):
            k = jac.shape[1]
            assert np.abs(out[12:21].reshape(3, 3)[:k].T - jac[0:3]).max() < 1e-14
            assert np.abs(out[21:30].reshape(3, 3)[:k].T - djac[0:3]).max() < 1e-14


@needs_lib
def test_device_exp_twist_on_host():
    lib = _capi.load()
    g = load_golden("g0_primitives.npz")
    for tw, H in zip(g["tw"], g["tw_exp"]):
        out = np.zeros(16)
        lib.arb_host_exp_twist(_capi._dp(np.ascontiguousarray(tw)), _capi._dp(out))
        assert np.abs(out.reshape(4, 4) - H).max() < 1e-12


@needs_lib
def test_device_zaligned_on_host():
    """homogeneousmatrix.py:201-232 incl. the argsort tie-breaking for axis-aligned normals."""
    lib = _capi.load()
    rng = np.random.default_rng(3)
    vecs = [v / np.linalg.norm(v) for v in rng.normal(size=(200, 3))]
    for a in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1],
              [1, 1, 0], [0, 1, 1], [1, 0, 1], [1, 1, 1], [-1, 1, -1], [1, -1, 0]):
        a = np.array(a, float)
        vecs.append(a / np.linalg.norm(a))
    for v in vecs:
        out = np.zeros(9)
        lib.arb_host_zaligned(_capi._dp(np.ascontiguousarray(v)), _capi._dp(out))
        assert np.abs(out.reshape(3, 3) - O.zaligned(v)[0:3, 0:3]).max() < 1e-15, v


@needs_lib
@pytest.mark.parametrize("geom", [0, 1, 2])
def test_device_narrow_phase_on_host(geom):
    """The kernel's narrow phase (compiled for the host) against the oracle's collisions.py restatement."""
    lib = _capi.load()
    rng = np.random.default_rng(10 + geom)
    half = np.array([0.5, 0.2, 0.8])
    plane = np.array([0., 0.6, 0.8, 0.1])
    n_inside = 0
    for k in range(300):
        H = np.eye(4)
        tw = rng.normal(size=6) * 0.7
        Hx = np.zeros(16)
        lib.arb_host_exp_twist(_capi._dp(tw), _capi._dp(Hx))
        H = Hx.reshape(4, 4)
        scale = 0.3 if (geom == 2 and k % 2) else 1.0         # half of the box cases start inside
        p = H[0:3, 0:3] @ (rng.uniform(-1, 1, 3) * scale * (half if geom == 2 and k % 2 else 1.)) + H[0:3, 3]
        rad, r0 = (0., 0.3) if k % 3 == 0 else (0.15, 0.25)
        gc0, gc1, Rc = np.zeros(3), np.zeros(3), np.zeros(9)
        sd = lib.arb_host_narrow_phase(geom, _capi._dp(np.ascontiguousarray(H.ravel())), _capi._dp(p), rad, r0,
                                       _capi._dp(half), _capi._dp(plane), _capi._dp(gc0), _capi._dp(gc1), _capi._dp(Rc))
        if geom == 0:
            esd, H0, H1 = O._plane_sphere_collision(H[None], plane, p[None], rad)
        elif geom == 1:
            esd, H0, H1 = O._sphere_sphere_collision(H[None, 0:3, 3], r0, p[None], rad)
        else:
            esd, H0, H1 = O._box_sphere_collision(H[None], half, p[None], rad)
            n_inside += esd[0] < -rad
        assert abs(sd - esd[0]) < 1e-13
        assert np.abs(gc0 - H0[0, 0:3, 3]).max() < 1e-13 and np.abs(gc1 - H1[0, 0:3, 3]).max() < 1e-13
        assert np.abs(Rc.reshape(3, 3) - H0[0, 0:3, 0:3]).max() < 1e-12
        assert np.abs(Rc.reshape(3, 3) - H1[0, 0:3, 0:3]).max() < 1e-12
    if geom == 2:
        assert n_inside > 50


@needs_lib
@pytest.mark.parametrize("dtype,tol", [(_capi.ARB_F64, 1e-9), (_capi.ARB_F32, 2e-3)])
def test_block_pinv_matches_numpy_pinv(dtype, tol):
    """The constraint blocks' (pseudo-)inverse as the kernels form it (arb_math.h inv_block + pinv_block, host build)
    against numpy.linalg.pinv, the call the reference makes (constraints.py:79, 83, 235, 795): regular blocks keep
    the pivoted elimination, rank-deficient ones (rank 1..nd-1, and the zero block) take the SVD route."""
    lib = _capi.load()
    rng = np.random.default_rng(3)
    for nd in (1, 2, 3, 4):
        for rank in range(0, nd + 1):
            for _ in range(6):
                A = rng.normal(size=(nd, rank)) @ rng.normal(size=(rank, nd)) if rank else np.zeros((nd, nd))
                A = np.ascontiguousarray(A * 10. ** rng.uniform(-3, 1))
                P = np.zeros((nd, nd))
                regular = lib.arb_host_block_pinv(dtype, nd, _capi._dp(A), _capi._dp(P))
                assert regular == (1 if rank == nd else 0), (nd, rank, regular)
                ref = np.linalg.pinv(A if dtype == _capi.ARB_F64 else A.astype(np.float32).astype(np.float64),
                                     rcond=1e-15 if dtype == _capi.ARB_F64 else 2e-5)
                scale = max(np.abs(ref).max(), 1e-300)
                if rank == nd:
                    # regular: conditioning of a random block times the arithmetic's epsilon
                    cond = np.linalg.cond(A)
                    assert np.abs(P - ref).max() / scale < max(tol, cond * (1e-15 if dtype == _capi.ARB_F64 else 2e-6)), (nd, cond)
                else:
                    assert np.abs(P - ref).max() / scale < tol, (nd, rank, np.abs(P - ref).max() / scale)
    # the blocks of the reference-generated singular scenarios (tests/golden/g12_singular.npz)
    g = load_golden("g12_singular.npz")
    assert g["loop_block_singular_values"][-1] < 1e-12 and g["contact_static_block_singular_values"][-1] < 1e-12


@needs_lib
def test_pivot_growth_measure_saturates_for_an_indefinite_pivot():
    """`zb - pb` on raw float bit patterns wrapped for a pivot <= 0 (Z_jj = 512, pivot = -1e-3 gave
    -1988301423: no warning exactly when the float32 elimination has gone indefinite).  arb_growth_bits (arb_math.h), the
    function phase C runs on the scalar unit, compiled for the host."""
    lib = _capi.load()
    g = lib.arb_host_growth_bits
    thr = 11 << 23                                  # the kernels warn above this
    INT_MAX = 0x7fffffff
    assert g(512., 512.) == 0
    assert abs(g(512., 1.) - (9 << 23)) <= 1        # 2^9: nine exponent steps
    assert g(512., 512. / 4096.) > thr > g(512., 512. / 1024.)
    for piv in (-1e-3, -0.0, 0.0, -512., float("nan"), float("inf"), -float("inf")):
        assert g(512., piv) == INT_MAX, piv         # pivot <= 0 or not finite: saturated, whatever Z_jj
    assert g(float("nan"), 1.) == INT_MAX
    assert g(-512., 1.) == g(512., 1.)              # a negative diagonal counts by its magnitude
    assert g(1e-3, 512.) < 0                        # a pivot larger than the diagonal: no growth


def test_design_register_table_is_generated_from_the_shipped_library():
    """DESIGN.md section 3 prints the registers / spills of the shipped kernels: the block is written by
    tools/gen_register_table.py from the library's code-object metadata (round 4's hand-written table had gone stale)."""
    import subprocess, sys
    tool = os.path.join(ROOT, "tools", "gen_register_table.py")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm tools not installed")
    r = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_python_constants_mirror_the_header():
    """Every `#define ARB_<NAME> <integer>` of include/arbstep.h that `_capi` repeats has the same value there (flags, status codes,
    warning bits, limits): the ctypes layer cannot see the C preprocessor."""
    import os
    import re
    from arboris_python_amd import _capi
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "arbstep.h")).read()
    defs = {k: int(v.rstrip("uU"), 0) for k, v in re.findall(r"#define\s+(ARB_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+u?|\d+u?)\b", header)}
    shared = [k for k in defs if hasattr(_capi, k)]
    assert len(shared) >= 12 and {"ARB_WIDE_MAX", "ARB_WIDE_MAX_CONSTRAINTS", "ARB_WARN_ACTIVE_CONSTRAINTS", "ARB_ABI_VERSION"} <= set(shared)
    for k in shared:
        assert getattr(_capi, k) == defs[k], (k, getattr(_capi, k), defs[k])
