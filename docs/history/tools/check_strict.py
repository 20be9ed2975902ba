#!/usr/bin/env python3
"""MLE fit modes against the CPU oracle: "strict" must reproduce the oracle's thetas and iteration counts bit for
bit; "refit" must do so on the re-fitted spots and stay within the north-star tolerance everywhere.
usage: python tools/check_strict.py [n_spots_per_box]"""
import json
import os
import sys
import time
from math import erf, sqrt

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def make_spots(box, n, seed, hard=False):
    rng = np.random.default_rng(seed)
    c = box // 2
    spots = np.empty((n, box, box), np.float32)
    idx = np.arange(box)
    for i in range(n):
        off = 1.5 if hard else 0.8
        x0, y0 = c + rng.uniform(-off, off), c + rng.uniform(-off, off)
        sx, sy = rng.uniform(0.6 if hard else 0.9, 0.25 * box + 0.3), rng.uniform(0.6 if hard else 0.9, 0.25 * box + 0.3)
        ex = np.array([0.5 * (erf((k - x0 + .5) / (sqrt(2) * sx)) - erf((k - x0 - .5) / (sqrt(2) * sx))) for k in idx])
        ey = np.array([0.5 * (erf((k - y0 + .5) / (sqrt(2) * sy)) - erf((k - y0 - .5) / (sqrt(2) * sy))) for k in idx])
        ph = rng.uniform(50, 9000) if hard else rng.uniform(1500, 9000)
        spots[i] = rng.poisson(ph * np.outer(ey, ex) + rng.uniform(0.5 if hard else 2, 30))
    return spots


def main():
    T = orc.max_threads()
    for box in (7, 5, 3, 9, 13, 15, 17, 21):
        for hard in (False, True):
            spots = make_spots(box, n, box + 100 * hard, hard)
            for method in ("sigmaxy", "sigma"):
                o = orc.gaussmle(spots, 1e-3, 100, method, threads=T)
                row = {"box": box, "hard": hard, "method": method, "n": n}
                for mode in ("strict", "refit", "fast"):
                    be.set_mle_mode(mode, 0.02)
                    t0 = time.perf_counter()
                    g = be.gaussmle_arrays(spots, 1e-3, 100, method)
                    dt = time.perf_counter() - t0
                    same_it = g[3] == o[3]
                    bit = np.all(g[0].view(np.uint32) == o[0].view(np.uint32), axis=1) | np.all(np.isnan(g[0]) == np.isnan(o[0]), axis=1) & np.all((g[0] == o[0]) | np.isnan(o[0]), axis=1)
                    below = o[3] < 100
                    dx = np.abs(g[0] - o[0])
                    with np.errstate(invalid="ignore"):
                        mx = float(np.nanmax(dx[below][:, [0, 1, 4, 5]])) if below.any() else 0.0
                    row[mode] = {"it_equal": int(same_it.sum()), "theta_bit_identical": int(bit.sum()),
                                 "max_dxysigma_below_max_it": mx, "ms": round(dt * 1e3, 1)}
                    if mode == "refit":
                        row[mode]["refit"] = be.last_refit_count()
                print(json.dumps(row), flush=True)
    be.set_mle_mode("refit", 0.02)


if __name__ == "__main__":
    main()
