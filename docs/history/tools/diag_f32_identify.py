#!/usr/bin/env python3
"""Which identifications differ between device and oracle on a float32 movie with NaN / inf pixels (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from picasso_amd import backend as be, _lib
box = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(300 + box)
F, Y, X = 7, 120, 336
mov = rng.poisson(35, size=(F, Y, X)).astype(np.float64)
for f in range(F):
    for _ in range(10):
        y, x = rng.integers(9, Y - 9), rng.integers(9, X - 9)
        s = rng.uniform(0.9, 1.0 + 0.15 * box)
        yy, xx = np.mgrid[y - 8:y + 9, x - 8:x + 9]
        mov[f, y - 8:y + 9, x - 8:x + 9] += rng.uniform(800, 5000) * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / s ** 2)
mov = mov + 0.5
for v in (np.nan, np.inf, -np.inf, -np.nan):
    for _ in range(12):
        mov[rng.integers(0, F), rng.integers(0, Y), rng.integers(0, X)] = v
mov = mov.astype(np.float32)
_lib.load().pmi_identify_set_narrow_chunk(3)
h = box // 2
for min_ng in (3000.0, 300.0, -1e9):
    for roi in (None, ((5, 11), (Y - 3, X - 13))):
        a = be.identify_arrays(mov, min_ng, box, roi=roi)
        b = orc.identify(mov, min_ng, box, roi=roi, threads=4)
        sa = set(zip(a[0].tolist(), a[1].tolist(), a[2].tolist())); sb = set(zip(b[0].tolist(), b[1].tolist(), b[2].tolist()))
        print("min_ng", min_ng, "roi", roi, "device", len(sa), "oracle", len(sb), "missing", len(sb - sa), "extra", len(sa - sb))
        for (f, y, x) in sorted(sb - sa)[:4]:
            k = [i for i in range(len(b[0])) if (b[0][i], b[1][i], b[2][i]) == (f, y, x)][0]
            w = mov[f, max(0, y - h - 1):y + h + 2, max(0, x - h - 1):x + h + 2]
            print("  missing", (f, y, x), "ng", b[3][k], "nonfinite in neighbourhood:", int((~np.isfinite(w)).sum()), "centre", mov[f, y, x])
        for (f, y, x) in sorted(sa - sb)[:4]:
            print("  extra", (f, y, x))
