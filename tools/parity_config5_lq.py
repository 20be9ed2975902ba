#!/usr/bin/env python3
"""Full-size parity of config 5 through the reference's 3-D DEFAULT route (picasso/zfit.py:300,472: fitting_method="gausslq"):
pmi_localize_lq_dev with box 13 on the 50 000-frame astigmatic movie, then the z fit, against the oracle's identify ->
get_spots -> lmdif -> table -> zfit on EVERY row (the strict mode is MINPACK's own arithmetic: the table columns are compared
for equality, z to the tolerance of the Brent search).  The movie (26 GB) stays on the device; the oracle follows it in
chunks of frames.  One JSON line.
usage: python tools/parity_config5_lq.py [frames] [chunk]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be, synth  # noqa: E402


def run(F=50000, chunk=2500, box=13):
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "zfit_calib3d.npz"))
    cx, cy = g["cx"], g["cy"]
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda", sigma=(1.1, 2.4), astigmatic=True,
                                 photons=(3000.0, 9000.0), seed=synth.DEFAULT_SEED + 5)
    torch.cuda.synchronize()
    assert be.get_lq_mode() == "strict"
    t = be.localize_lq_device(ctypes.c_void_p(movie.data_ptr()), np.uint16, (F, 512, 512), box, 5000.0, cam)
    second_pass, why = be.last_lq_refit_count(), be.last_lq_tie_reasons()
    z_gpu, _ = be.zfit_arrays(t["sx"], t["sy"], cx, cy)
    n = len(t["frame"])
    T = orc.max_threads()
    rows = id_mismatch = 0
    differ = {k: 0 for k in ("x", "y", "photons", "bg", "sx", "sy")}
    z_worst, z_nan_mismatch = 0.0, 0
    t0 = time.perf_counter()
    same = lambda a, b: (a == b) | (np.isnan(a) & np.isnan(b))      # noqa: E731
    for c0 in range(0, F, chunk):
        c1 = min(F, c0 + chunk)
        host = movie[c0:c1].cpu().numpy()
        fr, y, x, ng = orc.identify(host, 5000.0, box, threads=T)
        lo, hi = np.searchsorted(t["frame"], c0), np.searchsorted(t["frame"], c1)
        sl = slice(lo, hi)
        if hi - lo != len(fr) or not (np.array_equal(t["frame"][sl], (fr + c0).astype(np.uint32)) and np.array_equal(t["net_gradient"][sl], ng)):
            id_mismatch += 1
            continue
        spots = orc.get_spots(host, fr, y, x, box, cam)
        th = orc.gausslq(spots, threads=T)
        want = {"x": (th[:, 0].astype(np.float64) + x).astype(np.float32), "y": (th[:, 1].astype(np.float64) + y).astype(np.float32),
                "photons": th[:, 2], "bg": th[:, 3], "sx": th[:, 4], "sy": th[:, 5]}
        for k, w in want.items():
            differ[k] += int((~same(t[k][sl], w)).sum())
        oz, _ = orc.zfit(th[:, 4], th[:, 5], cx, cy, threads=T)
        ok = np.isfinite(oz)
        z_nan_mismatch += int((np.isfinite(z_gpu[sl]) != ok).sum())
        if ok.any():
            z_worst = max(z_worst, float(np.max(np.abs(z_gpu[sl][ok] - oz[ok]))))
        rows += len(fr)
    return {"workload": f"config 5 through gausslq: {F} frames x 512 x 512 uint16 astigmatic movie, box {box}, lmdif (strict mode), zfit",
            "localizations_gpu": int(n), "rows_compared": int(rows), "chunks_with_identification_mismatch": int(id_mismatch),
            "rows_not_bit_identical": differ, "z_max_abs_diff": z_worst, "z_finite_mismatch": int(z_nan_mismatch),
            "second_pass_spots": int(second_pass), "second_pass_reasons": why,
            "tolerance": {"table columns": "equality (NaN = NaN)", "z": 5e-5},
            "oracle_threads": T, "oracle_seconds": round(time.perf_counter() - t0, 1)}


if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    print(json.dumps(run(F, chunk)))
