#!/usr/bin/env python3
"""Which criterion sends how many spots of config 2's movie (and of an adversarial set) to the re-fit.
usage: python tools/mle_flag_breakdown.py [frames]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picasso_amd import backend as be, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
CAM = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
for method in ("sigmaxy", "sigma"):
    for eps, max_it in ((1e-3, 100), (1e-4, 100), (1e-2, 100), (1e-3, 5)):
        t = be.localize_mle_device(ctypes.c_void_p(movie.data_ptr()), np.uint16, tuple(movie.shape), 7, 5000.0, CAM,
                                   eps=eps, max_it=max_it, method=method)
        print(json.dumps({"workload": f"config 2, {F} frames", "method": method, "eps": eps, "max_it": max_it, "spots": len(t["frame"]),
                          "refit": be.last_refit_count(), "reasons": be.last_flag_reasons()}), flush=True)
