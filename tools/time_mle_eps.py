#!/usr/bin/env python3
"""Config 2 (pmi_localize_mle_dev, two ranges in flight) at the convergence thresholds the GUI offers: time per pass, share of
re-fitted spots and the criteria that sent them there.  usage: python tools/time_mle_eps.py [frames]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib, backend as be, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = _lib.load()
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 120 * F
tab = torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda")
dn = torch.zeros(1, dtype=torch.int64, device="cuda")


def run(eps):
    rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, 7, 5000.0, None, 0, F - 1, 100.0, 1.0, 1.0, eps, 100,
                                _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(tab.data_ptr()), cap, ctypes.c_void_p(dn.data_ptr()), None)
    _lib.check(rc, "localize")


base = None
for rep in range(2):
    for eps in (1e-3, 1e-4, 1e-2):
        for _ in range(3):
            run(eps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run(eps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        n = int(dn.item())
        if eps == 1e-3:
            base = dt
        print(json.dumps({"eps": eps, "ms_per_pass": round(dt * 1e3, 3), "vs_eps_1e-3": round(dt / base, 3), "localizations": n,
                          "refit": be.last_refit_count(), "refit_share": round(be.last_refit_count() / max(n, 1), 4),
                          "reasons": be.last_flag_reasons()}), flush=True)
