#!/usr/bin/env python3
"""Full-size parity of BASELINE.json config 2 (10 000 x 512 x 512, ~1e6 spots): every identification and every fit of
the fused device pipeline against the CPU oracle on the same movie.  One JSON line.
usage: python tools/parity_config2.py [frames]"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
mode = os.environ.get("PARITY_MODE", "refit")
margin = float(os.environ.get("PARITY_MARGIN", "0.001"))
be.set_mle_mode(mode, margin)
t = be.localize_mle_device(ctypes.c_void_p(movie.data_ptr()), np.uint16, (F, 512, 512), 7, 5000.0, cam)
refit = be.last_refit_count()
host = movie.cpu().numpy()
T = orc.max_threads()
fr, y, x, ng = orc.identify(host, 5000.0, 7, threads=T)
spots = orc.get_spots(host, fr, y, x, 7, cam)
th, cr, ll, it, close = orc.gaussmle_closeness(spots, 1e-3, 100, "sigmaxy", threads=T)
n = len(fr)
same = t["iterations"] == it
d = {"x": np.abs(t["x"] - (th[:, 0] + x - 3)), "y": np.abs(t["y"] - (th[:, 1] + y - 3)),
     "sx": np.abs(t["sx"] - th[:, 4]), "sy": np.abs(t["sy"] - th[:, 5]),
     "photons_rel": np.abs(t["photons"] - th[:, 2]) / th[:, 2], "bg": np.abs(t["bg"] - th[:, 3]),
     "lpx_rel": np.abs(t["lpx"] - np.sqrt(cr[:, 0])) / np.sqrt(cr[:, 0]),
     "log_likelihood_rel": np.abs(t["log_likelihood"] - ll) / np.abs(ll)}
below = it < 100
diff_it = np.flatnonzero(~same)
out = {"frames": F, "mle_mode": mode, "refit_margin": margin, "refit_spots": refit, "refit_frac": refit / max(n, 1),
       "iteration_histogram_oracle": {str(k): int(v) for k, v in zip(*np.unique(np.minimum(it, 100) // 10 * 10, return_counts=True))},
       "rows_with_different_iterations": [{"row": int(i), "gpu_it": int(t["iterations"][i]), "oracle_it": int(it[i]),
                                           "dx": float(d_) , "dsx": float(e_)} for i, d_, e_ in
                                          zip(diff_it[:20], np.abs(t["x"] - (th[:, 0] + x - 3))[diff_it[:20]],
                                              np.abs(t["sx"] - th[:, 4])[diff_it[:20]])],
       "n_rows_with_different_iterations": int(len(diff_it)),
       "oracle_decision_margin_of_differing_rows": {"max": float(close[diff_it].max()) if len(diff_it) else 0.0,
                                                    "sorted_top": [float(v) for v in np.sort(close[diff_it])[::-1][:12]]},
       "oracle_decision_margin_quantiles_all_rows": {q: float(np.quantile(close, float(q))) for q in ("0.001", "0.01", "0.05", "0.1", "0.5")}, "identifications_gpu": int(len(t["frame"])), "identifications_oracle": int(n),
       "identification_rows_identical": bool(len(t["frame"]) == n and np.array_equal(t["frame"], fr)
                                             and np.array_equal(t["net_gradient"], ng)),
       "same_iteration_count_frac": float(same.mean()),
       "iteration_count_differs_by_more_than_1": int((np.abs(t["iterations"].astype(np.int64) - it) > 1).sum()),
       "bit_identical_theta_frac": float(np.mean((t["photons"] == th[:, 2]) & (t["sx"] == th[:, 4]) & (t["bg"] == th[:, 3]))),
       "max_abs_diff_all_spots": {k: float(np.nanmax(v)) for k, v in d.items()},
       "max_abs_diff_same_iterations": {k: float(np.nanmax(v[same])) for k, v in d.items()},
       "max_abs_diff_rows_below_max_it": {k: float(np.nanmax(v[below])) for k, v in d.items()},
       "rows_at_max_it_oracle": int((~below).sum()),
       "tolerance": {"x,y,sigma": 1e-3, "photons_rel": 1e-2}, "oracle_threads": T}
print(json.dumps(out))
