#!/usr/bin/env python3
"""A/B on one box: pmi_localize_mle_dev with identify's exact stage in the scan (defer 0) and in the fit's start-value
kernel (defer 1): the two tables must be equal bit for bit; time per pass for one and two ranges in flight.
usage: python tools/ab_defer.py [frames] [box] [reps] [only] [eps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
box = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
only = int(sys.argv[4]) if len(sys.argv) > 4 else -1          # profile one mode: 0 / 1 (one range in flight, 10 passes)
EPS = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3
L = _lib.load()
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 130 * F
tabs = {d: torch.zeros((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda") for d in (0, 1)}
dn = torch.zeros(1, dtype=torch.int64, device="cuda")


def run(defer):
    rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, box, 5000.0, None, 0, F - 1, 100.0, 1.0, 1.0, EPS, 100,
                                _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(tabs[defer].data_ptr()), cap, ctypes.c_void_p(dn.data_ptr()), None)
    _lib.check(rc, "localize")


if only >= 0:
    _lib.check(L.pmi_localize_set_ranges(1), "ranges")
    _lib.check(L.pmi_localize_set_defer(only), "defer")
    for _ in range(10):
        run(only)
    torch.cuda.synchronize()
    print("rows", int(dn.item()))
    sys.exit(0)
counts = {}
for ranges in (1, 2):
    _lib.check(L.pmi_localize_set_ranges(ranges), "ranges")
    for defer in (0, 1):
        _lib.check(L.pmi_localize_set_defer(defer), "defer")
        tabs[defer].zero_()
        run(defer)
        torch.cuda.synchronize()
        counts[defer] = int(dn.item())
    n = counts[0]
    same = counts[0] == counts[1] and bool(torch.equal(tabs[0][:, :n], tabs[1][:, :n]))
    print(f"ranges {ranges}: rows {counts[0]} / {counts[1]}, tables identical: {same}", flush=True)
    if not same and counts[0] == counts[1]:
        diff = (tabs[0][:, :n] != tabs[1][:, :n]).nonzero()
        print("  first differences (column, row):", diff[:8].tolist(), flush=True)
for rep in range(reps):
    for ranges in (1, 2):
        _lib.check(L.pmi_localize_set_ranges(ranges), "ranges")
        for defer in (0, 1):
            _lib.check(L.pmi_localize_set_defer(defer), "defer")
            for _ in range(3):
                run(defer)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run(defer)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f"ranges {ranges} defer {defer}: {dt * 1e3:.3f} ms per pass, {int(dn.item())} localizations", flush=True)
_lib.check(L.pmi_localize_set_defer(1), "defer")
_lib.check(L.pmi_localize_set_ranges(2), "ranges")
