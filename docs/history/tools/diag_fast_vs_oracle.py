#!/usr/bin/env python3
"""Diagnostic: where does the float32 Newton loop ("fast" mode) leave the oracle beyond the tolerance, and would the
borderline flag have caught it?  Dumps the offending spots to gpurun_out/diag_fast.npz.
usage: python tools/diag_fast_vs_oracle.py [n] [box]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_strict import make_spots  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
box = int(sys.argv[2]) if len(sys.argv) > 2 else 7
MODE = os.environ.get("DIAG_MODE", "fast")
MARGIN = float(os.environ.get("DIAG_MARGIN", "0.001"))
T = orc.max_threads()
dump = {}
for hard in (True, False):
    spots = make_spots(box, n, 1234 + hard, hard)
    for method in ("sigmaxy", "sigma"):
        o = orc.gaussmle_closeness(spots, 1e-3, 100, method, threads=T)
        be.set_mle_mode(MODE, MARGIN)
        g = be.gaussmle_arrays(spots, 1e-3, 100, method)
        refit = be.last_refit_count()
        below = (o[3] < 100) & (g[3] < 100)
        d = np.abs(g[0] - o[0])
        with np.errstate(invalid="ignore"):
            bad = below & ((d[:, [0, 1, 4, 5]].max(axis=1) > 1e-3) | (d[:, 2] / np.maximum(np.abs(o[0][:, 2]), 1) > 1e-2))
        badrows = np.flatnonzero(bad)
        caught = (o[4][badrows] < MARGIN) if MODE == "fast" else np.zeros(len(badrows), bool)
        row = {"box": box, "hard": hard, "method": method, "mode": MODE, "margin": MARGIN, "n": n, "refit": refit, "bad": int(bad.sum()),
               "bad_caught_by_margin": int(caught.sum()),
               "it_differs": int((g[3] != o[3]).sum()), "hit_max_it_oracle": int((o[3] >= 100).sum()), "hit_max_it_gpu": int((g[3] >= 100).sum()),
               "uncaught_examples": [{"row": int(r), "it_gpu": int(g[3][r]), "it_orc": int(o[3][r]), "close": float(o[4][r]),
                                      "theta_gpu": [round(float(v), 4) for v in g[0][r]], "theta_orc": [round(float(v), 4) for v in o[0][r]]}
                                     for r in badrows[~caught][:8]]}
        print(json.dumps(row), flush=True)
        key = f"{'hard' if hard else 'easy'}_{method}"
        unc = badrows[~caught][:2000]
        dump[key + "_spots"] = spots[unc]
        dump[key + "_theta_gpu"] = g[0][unc]; dump[key + "_theta_orc"] = o[0][unc]
        dump[key + "_it_gpu"] = g[3][unc]; dump[key + "_it_orc"] = o[3][unc]
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/diag_{MODE}_{box}.npz", **dump)
