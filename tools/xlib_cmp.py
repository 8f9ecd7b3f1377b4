#!/usr/bin/env python3
"""Bitwise comparison of two tools/xlib_dump.py outputs."""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = k in b.files and a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
    if not same:
        bad += 1
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)) if k in b.files else None
        print("DIFF", k, None if d is None else (float(np.nanmax(d)), int((d > 0).sum()), d.size))
print("%d arrays, %d differ" % (len(a.files), bad))
sys.exit(1 if bad else 0)
