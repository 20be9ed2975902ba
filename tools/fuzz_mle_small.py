#!/usr/bin/env python3
"""Focused differential run of the MLE fit on the hard corner of tools/fuzz_parity.py: small boxes, spots far off
centre, negative pixels, both eps.  Every row must be on the oracle's iteration count and, where it converged,
within 1e-3 px.  Prints the offending spots' fitted background / photons (are they on a clamp?).
usage: python tools/fuzz_mle_small.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from picasso_amd import backend as be

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t_end = time.time() + budget
cases = bad_rows = rows = 0
while time.time() < t_end:
    box = int(rng.choice([3, 3, 5, 5, 7, 9, 15]))
    n = 512
    c = box // 2
    idx = np.arange(box)
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        x0, y0 = c + rng.uniform(-1.5, 1.5, 2)
        sx, sy = rng.uniform(0.5, 0.3 * box + 0.5, 2)
        gx = np.exp(-0.5 * ((idx - x0) / sx) ** 2) / (np.sqrt(2 * np.pi) * sx)
        gy = np.exp(-0.5 * ((idx - y0) / sy) ** 2) / (np.sqrt(2 * np.pi) * sy)
        spots[i] = rng.poisson(rng.uniform(20, 9000) * np.outer(gy, gx) + rng.uniform(0.05, 60))
    spots -= np.float32(rng.choice([0.0, 0.0, 3.0]))
    method = ["sigmaxy", "sigma"][rng.integers(0, 2)]
    eps = float(rng.choice([1e-3, 1e-3, 1e-2]))
    max_it = 100
    th, cr, ll, it = be.gaussmle_arrays(spots, eps, max_it, method)
    oth, ocr, oll, oit = orc.gaussmle(spots, eps, max_it, method, threads=8)
    cases += 1; rows += n
    fin = np.all(np.isfinite(oth), axis=1) & (oit < max_it)
    d = np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1)
    bad = (it != oit) | (fin & (d > 1e-3))
    for w in np.flatnonzero(bad):
        bad_rows += 1
        print(f"box {box} {method} eps {eps}: it {it[w]} / {oit[w]}  d {d[w]:.2e}  oracle N {oth[w, 2]:.4g} bg {oth[w, 3]:.5g} s {oth[w, 4]:.4g} {oth[w, 5]:.4g}"
              f"  gpu N {th[w, 2]:.4g} bg {th[w, 3]:.5g}  spot min {spots[w].min():.3g} sum {spots[w].sum():.5g}", flush=True)
print(f"done: {cases} cases, {rows} rows, {bad_rows} rows off")
