#!/usr/bin/env python3
"""Scan and fit time of pmi_localize_mle_dev on config 2 (library-side HIP events, one frame range).  usage: python tools/time_stages.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from picasso_amd import _lib, synth
L = _lib.load()
F = 10000
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 120 * F
tab = torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda")
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
def run():
    _lib.check(L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, 7, 5000.0, None, 0, F - 1, 100.0, 1.0, 1.0, 1e-3, 100,
                                      _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(tab.data_ptr()), cap, ctypes.c_void_p(dn.data_ptr()), None), "loc")
for _ in range(3): run()
torch.cuda.synchronize()
L.pmi_set_kernel_timing(1)
a, b = ctypes.c_float(0), ctypes.c_float(0)
sa, sb = [], []
for _ in range(8):
    run(); torch.cuda.synchronize(); L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b)); sa.append(a.value); sb.append(b.value)
print(os.environ.get("PMI_MLE_NO_HANDOFF", "handoff"), "scan %.3f ms fit %.3f ms" % (np.median(sa), np.median(sb)), int(dn.item()), "rows", flush=True)
