#!/usr/bin/env python3
"""Identify on config 2's movie with a saturated square (a fiducial) and a second plateau in every frame: time and a
checksum of the result (run once plain, once with PMI_IDENTIFY_GENERIC=1: the checksums must agree).
usage: python tools/time_identify_saturated.py [frames] [side]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from picasso_amd import _lib, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
side = int(sys.argv[2]) if len(sys.argv) > 2 else 48
L = _lib.load()
mov = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
mov[:, 100:100 + side, 200:200 + side] = 65535
mov[:, 300:300 + side // 2, 40:40 + 3 * side] = 40000          # a second plateau below saturation
torch.cuda.synchronize()
cap = 400 * F
out = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(3)] + [torch.empty(cap, dtype=torch.float32, device="cuda")]
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
L.pmi_set_kernel_timing(1)
a, b = ctypes.c_float(0), ctypes.c_float(0)
import hashlib
ts = []
for _ in range(3):
    _lib.check(L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()), 0, F, 512, 512, 7, 5000.0, None, 0, F - 1,
                                  *[ctypes.c_void_p(t.data_ptr()) for t in out], cap, ctypes.c_void_p(dn.data_ptr()), None))
    torch.cuda.synchronize()
    L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
    ts.append(a.value)
n = int(dn.item())
h = hashlib.sha1()
for t in out:
    h.update(t[:n].cpu().numpy().tobytes())
label = "generic" if os.environ.get("PMI_IDENTIFY_GENERIC") else "fast"
print(f"{label:8s} n={n:8d} scan {min(ts[1:]):9.3f} ms  {mov.numel() * 2 / (min(ts[1:]) * 1e-3) / 1e9:8.1f} GB/s  sha1 {h.hexdigest()[:16]}")
