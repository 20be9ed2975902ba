#!/usr/bin/env python3
"""Times RCC (imageprocess.rcc: pairwise correlations + peak fits) on synthetic segment images.
usage: python tools/time_rcc.py [n_segments] [size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picasso_amd import backend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rng = np.random.default_rng(0)
pts = rng.uniform(20, size - 20, (4000, 2))
seg = np.zeros((n, size, size))
yy, xx = np.mgrid[-4:5, -4:5]
blob = np.exp(-0.5 * (yy ** 2 + xx ** 2) / 1.5 ** 2)
for i in range(n):
    d = rng.normal(0, 0.7, 2) + 0.05 * i
    for py, px in pts[rng.random(len(pts)) < 0.8]:
        y0, x0 = int(py + d[0]), int(px + d[1])
        seg[i, y0 - 4:y0 + 5, x0 - 4:x0 + 5] += blob
for rep in range(3):
    t0 = time.perf_counter()
    shifts, status = backend.rcc_shifts_arrays(seg, 32, 5)
    dt = time.perf_counter() - t0
    print(f"{n} segments of {size}^2: {len(status)} pairs in {dt * 1e3:.1f} ms; fit status counts {np.unique(status, return_counts=True)}")
