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
