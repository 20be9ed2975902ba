#!/usr/bin/env python3
"""Prototype: the fit of frame range A beside the scan of frame range B (two streams, two scratch banks) against the
same two ranges one after the other on one stream.  usage: python tools/proto_overlap.py [frames] [parts]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = _lib.load()
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
per = F // parts
cap = 120 * per
tabs = [torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda") for _ in range(parts)]
dns = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(2)]
px = 512 * 512 * 2


def run(k, stream, bank):
    _lib.check(L.pmi_scratch_bank(bank), "bank")
    rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr() + k * per * px), 0, per, 512, 512, 7, 5000.0, None, 0, per - 1,
                                100.0, 1.0, 1.0, 1e-3, 100, _lib.MLE_METHODS["sigmaxy"],
                                ctypes.c_void_p(tabs[k].data_ptr()), cap, ctypes.c_void_p(dns[k].data_ptr()),
                                ctypes.c_void_p(stream.cuda_stream))
    _lib.check(rc, "localize")


def serial():
    for k in range(parts):
        run(k, streams[0], 0)


def overlapped():
    for k in range(parts):
        run(k, streams[k % 2], k % 2)


for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    n = sum(int(d.item()) for d in dns)
    print(f"{name:11s} {parts} parts of {per} frames: {dt * 1e3:.3f} ms per pass, {n} localizations, {n / dt:.3e} /s")
_lib.check(L.pmi_scratch_bank(0), "bank")
