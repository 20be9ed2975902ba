#!/usr/bin/env python3
"""gausslq device vs oracle on the fuzz distribution: bit-identical fraction, re-fitted (tie) spots, worst differences.
usage: python tools/diag_lq.py [n] [boxes]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
boxes = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [7, 3, 9, 13]


def lq_spots(box, n, seed):
    rng = np.random.default_rng(seed)
    c = box // 2
    idx = np.arange(box) - c
    x0 = rng.uniform(-1.2, 1.2, n); y0 = rng.uniform(-1.2, 1.2, n)
    sx = rng.uniform(0.6, 0.25 * box + 0.5, n); sy = rng.uniform(0.6, 0.25 * box + 0.5, n)
    gx = np.exp(-0.5 * ((idx[None] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
    gy = np.exp(-0.5 * ((idx[None] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
    return rng.poisson(rng.uniform(100, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :] + rng.uniform(0.5, 60, n)[:, None, None]).astype(np.float32)


for box in boxes:
    spots = lq_spots(box, n, 40 + box)
    if os.environ.get("DIAG_REAL"):          # the widths and photon counts of config 3's movie instead of the fuzz distribution
        rng = np.random.default_rng(7 + box)
        idx = np.arange(box) - box // 2
        x0 = rng.uniform(-0.6, 0.6, n); y0 = rng.uniform(-0.6, 0.6, n); sg = rng.uniform(0.9, 1.4, n)
        gx = np.exp(-0.5 * ((idx[None] - x0[:, None]) / sg[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sg[:, None])
        gy = np.exp(-0.5 * ((idx[None] - y0[:, None]) / sg[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sg[:, None])
        spots = rng.poisson(rng.uniform(2000, 8000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :] + rng.uniform(10, 30, n)[:, None, None]).astype(np.float32)
    th, info, nfev = be.gausslq_arrays(spots, full_output=True)
    refit = be.last_lq_refit_count()
    print("tie reasons", be.last_lq_tie_reasons(), flush=True)
    oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=orc.max_threads())
    exact = np.all((th == oth) | (np.isnan(th) & np.isnan(oth)), axis=1)
    fin = np.all(np.isfinite(oth), axis=1) & np.all(np.isfinite(th), axis=1)
    d = np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1)
    d[~fin] = 0
    print(json.dumps({"box": box, "n": n, "refit": refit, "bit_identical": float(exact.mean()), "not_identical": int((~exact).sum()),
                      "info_differs": int((info != oinfo).sum()), "nfev_differs": int((nfev != onfev).sum()),
                      "beyond_1e-3": int((d > 1e-3).sum()), "worst_px": float(d.max()),
                      "nonfinite_mismatch": int((np.isfinite(th).all(axis=1) != np.isfinite(oth).all(axis=1)).sum())}), flush=True)
    bad = np.flatnonzero(~exact)[:5]
    for r in bad:
        print("   row", int(r), "info", int(info[r]), int(oinfo[r]), "nfev", int(nfev[r]), int(onfev[r]), "d", float(d[r]),
              "gpu", np.round(th[r], 5).tolist(), "orc", np.round(oth[r], 5).tolist(), flush=True)
