import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def load_model(name):
    from arboris_python_amd.flatten import FlatModel
    d = np.load(os.path.join(GOLDEN, "model_%s.npz" % name))
    m = FlatModel.from_npz_dict({k: d[k] for k in d.files if k not in ("q0", "dq0")})
    return m, d["q0"], d["dq0"]


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def model():
    return load_model


def oracle_sensitivity(m, q, dq, dt, samples=4, seed=0, **kw):
    """How far the float64 oracle's own (q+, dq+) move when its input moves by one float32 ulp (relative 2^-24,
    random signs): the conditioning of the step as ANY float32 arithmetic sees it.  Per-world relative changes."""
    import arb_oracle as O
    rng = np.random.default_rng(seed)
    q, dq = np.asarray(q, np.float64), np.asarray(dq, np.float64)
    bq, bdq, _ = O.step(m, q, dq, dt, **kw)
    sq, sdq = np.zeros(len(q)), np.zeros(len(q))
    for _ in range(samples):
        pq = q * (1. + 2. ** -24 * rng.choice([-1., 1.], q.shape))
        pdq = dq * (1. + 2. ** -24 * rng.choice([-1., 1.], dq.shape))
        oq, odq, _ = O.step(m, pq, pdq, dt, **kw)
        sq = np.maximum(sq, np.abs(oq - bq).max(axis=1) / np.maximum(1., np.abs(bq).max(axis=1)))
        sdq = np.maximum(sdq, np.abs(odq - bdq).max(axis=1) / np.maximum(1., np.abs(bdq).max(axis=1)))
    return sq, sdq


def assert_f32_parity(m, q_in, dq_in, dt, got_q, got_dq, ref_q, ref_dq, tol=1e-5, cap=3e-5, **kw):
    """Per-world float32 gate: max|x - x_ref| / max(1, max|x_ref|) < tol for q+ and dq+.  A world over the gate is
    accepted only when the step itself is that ill-conditioned -- the float64 oracle moves by at least half the
    observed error under a one-ulp (float32) change of its input -- and the error stays below `cap`."""
    got_q, got_dq = np.asarray(got_q, np.float64), np.asarray(got_dq, np.float64)
    eq = np.abs(got_q - ref_q).max(axis=1) / np.maximum(1., np.abs(ref_q).max(axis=1))
    edq = np.abs(got_dq - ref_dq).max(axis=1) / np.maximum(1., np.abs(ref_dq).max(axis=1))
    over = np.flatnonzero((eq >= tol) | (edq >= tol))
    if len(over):
        f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        kw_over = {k: (np.asarray(v)[over] if hasattr(v, "__len__") and len(v) == len(eq) else v) for k, v in kw.items()}
        sq, sdq = oracle_sensitivity(m, f32(q_in)[over], f32(dq_in)[over], dt, **kw_over)
        for k, w in enumerate(over):
            assert max(eq[w], edq[w]) < cap and eq[w] <= max(tol, 2 * sq[k]) and edq[w] <= max(tol, 2 * sdq[k]), \
                "world %d: err q %.2e dq %.2e, oracle sensitivity q %.2e dq %.2e" % (w, eq[w], edq[w], sq[k], sdq[k])
            print("world %d over the %.0e gate (q %.2e dq %.2e): ill-conditioned step, one float32 ulp on the input moves "
                  "the oracle by q %.1e dq %.1e" % (w, tol, eq[w], edq[w], sq[k], sdq[k]))
    return float(eq.max()), float(edq.max())
