"""Bootstrap for importing the upstream reference (THIS container only).

The reference at /root/reference is Python-2-era code.  It imports under
Python 3 / NumPy 2 once four tiny compatibility shims are installed (SURVEY.md
section 8c).  This module is used only by the golden-vector generator
(tools/gen_golden.py) and by tests that are skipped when /root/reference is
absent; nothing here travels to the GPU box in any useful form because the
reference itself does not exist there.
"""
import builtins
import itertools
import os
import sys

REFERENCE_ROOT = os.environ.get("ARBORIS_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "arboris"))


def load():
    """Import the reference `arboris` package with the py3 shims; return it."""
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    builtins.unicode = str                       # robots/human36.py:106
    itertools.imap = map                         # core.py:1081
    import numpy as np
    import arboris
    if not arboris.__file__.startswith(REFERENCE_ROOT):
        raise RuntimeError("`arboris` resolved to %s, not the reference"
                           % arboris.__file__)
    import arboris.controllers as _c            # controllers.py:37: py3 filter() is one-shot
    _c.filter = lambda f, it: list(builtins.filter(f, it))
    import arboris.homogeneousmatrix as _Hg     # homogeneousmatrix.py:309-316: ragged list

    def _adjoint(H):
        assert _Hg.ishomogeneousmatrix(H), H
        R, p = H[0:3, 0:3], H[0:3, 3]
        px = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
        return np.vstack((np.hstack((R, np.zeros((3, 3)))),
                          np.hstack((np.dot(px, R), R))))
    _Hg.adjoint = _adjoint
    import arboris.core
    import arboris.joints
    import arboris.constraints
    import arboris.collisions
    import arboris.shapes
    import arboris.massmatrix
    import arboris.twistvector
    import arboris.rigidmotion
    import arboris.robots.simplearm
    import arboris.robots.snake
    import arboris.robots.human36
    import arboris.robots.simpleshapes
    # human36.py does `from arboris.homogeneousmatrix import adjoint` at import
    # time, binding the unpatched function; rebind it.
    arboris.robots.human36.adjoint = _adjoint
    return arboris
