#!/usr/bin/env python3
"""A/B on one box: pmi_localize_mle_dev with one frame range and with two in flight.  usage: python tools/time_ranges.py [frames]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = _lib.load()
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 120 * F
tab = torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda")
dn = torch.zeros(1, dtype=torch.int64, device="cuda")


def run():
    rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, 7, 5000.0, None, 0, F - 1, 100.0, 1.0, 1.0, 1e-3, 100,
                                _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(tab.data_ptr()), cap, ctypes.c_void_p(dn.data_ptr()), None)
    _lib.check(rc, "localize")


for rep in range(3):
    for ranges in (1, 2):
        _lib.check(L.pmi_localize_set_ranges(ranges), "ranges")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"ranges {ranges}: {dt * 1e3:.3f} ms per pass, {int(dn.item())} localizations", flush=True)
