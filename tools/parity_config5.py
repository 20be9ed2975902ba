#!/usr/bin/env python3
"""Full-size parity of BASELINE.json config 5 (13x13 ROI astigmatic MLE + zfit, ~5e6 spots): every identification, every fit
and every z of the fused device pipeline (pmi_localize_mle_dev with box 13, then pmi_zfit) against the CPU oracle on the
same 50 000-frame movie.  The movie (26 GB) stays on the device; the oracle follows it in chunks of frames.  One JSON line.
usage: python tools/parity_config5.py [frames] [chunk]"""
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
    t = be.localize_mle_device(ctypes.c_void_p(movie.data_ptr()), np.uint16, (F, 512, 512), box, 5000.0, cam)
    refit, why = be.last_refit_count(), be.last_flag_reasons()
    z_gpu, dz_gpu = be.zfit_arrays(t["sx"], t["sy"], cx, cy)
    n = len(t["frame"])
    T = orc.max_threads()
    h = box // 2
    worst = {k: 0.0 for k in ("x", "y", "sx", "sy", "photons_rel", "bg", "lpx_rel", "z", "d_zcalib")}
    rows = it_differ = id_mismatch = at_max_it = 0
    t0 = time.perf_counter()
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
        th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=T)
        oz, odz = orc.zfit(th[:, 4], th[:, 5], cx, cy, threads=T)
        conv = it < 100
        at_max_it += int((~conv).sum())
        it_differ += int((t["iterations"][sl] != it).sum())
        d = {"x": np.abs(t["x"][sl] - (th[:, 0] + x - h)), "y": np.abs(t["y"][sl] - (th[:, 1] + y - h)),
             "sx": np.abs(t["sx"][sl] - th[:, 4]), "sy": np.abs(t["sy"][sl] - th[:, 5]),
             "photons_rel": np.abs(t["photons"][sl] - th[:, 2]) / np.maximum(th[:, 2], 1), "bg": np.abs(t["bg"][sl] - th[:, 3]),
             "lpx_rel": np.abs(t["lpx"][sl] - np.sqrt(cr[:, 0])) / np.sqrt(cr[:, 0]),
             "z": np.abs(z_gpu[sl] - oz), "d_zcalib": np.abs(dz_gpu[sl] - odz)}
        for k, v in d.items():
            v = v[conv]
            if len(v):
                worst[k] = max(worst[k], float(np.nanmax(v)))
        rows += len(fr)
    return {"workload": f"config 5: {F} frames x 512 x 512 uint16 astigmatic movie, box {box}, MLE sigmaxy eps 1e-3 max_it 100, zfit",
            "localizations_gpu": int(n), "rows_compared": int(rows), "chunks_with_identification_mismatch": int(id_mismatch),
            "rows_with_different_iterations": int(it_differ), "rows_at_max_it_oracle": int(at_max_it),
            "max_abs_diff_rows_below_max_it": worst, "refit_spots": int(refit), "flag_reasons": why,
            "tolerance": {"x,y,sigma": 1e-3, "photons_rel": 1e-2, "z": "0.05 nm (the widths agree to 2e-6 px; the calibration turns 1 px into up to 1e4 nm)"},
            "oracle_threads": T, "oracle_seconds": round(time.perf_counter() - t0, 1)}


if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    print(json.dumps(run(F, chunk)))
