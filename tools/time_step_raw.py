#!/usr/bin/env python3
"""Config 2's step through the bare C ABI of ONE library file (A/B of two builds on one box: a process each, alternating).
usage: python tools/time_step_raw.py <libpicasso_hip.so> [steps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import synth

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
F = 10000
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 120 * F
tab = torch.empty((17, cap), dtype=torch.int32, device="cuda")
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
f = lib.pmi_localize_mle_dev
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
              ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
              ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]


def run():
    rc = f(movie.data_ptr(), 0, F, 512, 512, 7, 5000.0, None, 0, F - 1, 100.0, 1.0, 1.0, 1e-3, 100, 1, tab.data_ptr(), cap, dn.data_ptr(), None)
    assert rc == 0, rc


for _ in range(5):
    run()
torch.cuda.synchronize()
out = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    out.append(1e3 * (time.perf_counter() - t0) / steps)
print(os.path.basename(sys.argv[1]), " ".join(f"{v:.4f}" for v in out), "ms per step,", int(dn.item()), "localizations")
