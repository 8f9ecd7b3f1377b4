"""Device unit test of SoftFingerContact.solve as the kernels run it (arb_math.h compiled for gfx950, one lane
per input tuple, lane-private LDS work array for the eig6 fallback) against the same code compiled for the
host and against the reference tuples recorded in tests/golden/g3_contacts.npz, plus the solves of the
eig6-fallback canary state (tests/golden/canary_eig6_fallback.npz)."""
import ctypes as C

import numpy as np
import pytest

import arb_oracle as O
from conftest import load_golden, load_model
from arboris_python_amd import _capi

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def tuples_of_trace(tr):
    return np.array([np.concatenate([t["vel"], np.asarray(t["adm"]).ravel(), t["force"], [t["sdist"], t["dt"], t["mu"]]])
                     for t in tr])


def host_solve(lib, dtype, tup):
    vel, adm, force = tup[0:4].copy(), tup[4:20].copy(), tup[20:24].copy()
    df, eps = np.zeros(4), np.ones(3)
    br = lib.arb_host_softfinger_solve(dtype, _capi._dp(vel), _capi._dp(adm), _capi._dp(force), float(tup[24]), float(tup[25]),
                                       float(tup[26]), _capi._dp(eps), _capi._dp(df))
    return np.concatenate([force, df, [br]])


def device_solve(lib, dtype, tuples):
    tin = np.ascontiguousarray(tuples, dtype=np.float64)
    out = np.zeros((len(tin), 9))
    _capi.check(lib.arb_dev_softfinger_solve(dtype, 0, len(tin), _capi._dp(tin), _capi._dp(out)))
    return out


@pytest.fixture(scope="module")
def canary_tuples():
    m, _, _ = load_model("human36_c4")
    d = load_golden("canary_eig6_fallback.npz")
    tr = []
    O.step(m, d["q"].astype(np.float64), d["dq"].astype(np.float64), 5e-3, trace=tr)
    assert len(tr) == 80
    return tuples_of_trace(tr), np.array([np.concatenate([t["force"] + t["dforce"], t["dforce"], [t["branch"]]]) for t in tr])


@pytest.mark.parametrize("dtype,tol", [(_capi.ARB_F64, 1e-9), (_capi.ARB_F32, 2e-4)])
@pytest.mark.parametrize("route", [0, 0x100])            # fast sliding shift / forced eig6 fallback
def test_device_solve_equals_host_and_oracle(canary_tuples, dtype, tol, route):
    assert torch.cuda.is_available()
    lib = _capi.load()
    tuples, ref = canary_tuples
    dev = device_solve(lib, dtype | route, tuples)
    host = np.array([host_solve(lib, dtype | route, t) for t in tuples])
    scale = max(1., np.abs(ref[:, :4]).max())
    assert np.isfinite(dev).all()
    assert np.array_equal(dev[:, 8], host[:, 8]) and np.array_equal(np.minimum(dev[:, 8], 2), ref[:, 8])
    assert np.abs(dev[:, :8] - host[:, :8]).max() / scale < tol
    assert np.abs(dev[:, :8] - ref[:, :8]).max() / scale < tol
