#!/usr/bin/env python3
"""Identify on a float32 copy of config 2's movie: integer counts in float32 (narrowed exactly, packed scan), or — `frac` —
the same movie scaled to fractions (16-bit keys, packed scan, exact decisions on the float32 pixels).
usage: python tools/time_identify_f32.py [frames] [frac]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = _lib.load()
mov = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda").view(torch.int16).to(torch.float32)
frac = len(sys.argv) > 2 and sys.argv[2] == "frac"
if frac:
    mov = mov * 1.37 + 0.25
torch.cuda.synchronize()
cap = 400 * F
out = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(3)] + [torch.empty(cap, dtype=torch.float32, device="cuda")]
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
L.pmi_set_kernel_timing(1)
a, b = ctypes.c_float(0), ctypes.c_float(0)
ts = []
for _ in range(4):
    _lib.check(L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()), 5, F, 512, 512, 7, 5000.0, None, 0, F - 1,
                                  *[ctypes.c_void_p(t.data_ptr()) for t in out], cap, ctypes.c_void_p(dn.data_ptr()), None))
    torch.cuda.synchronize()
    L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
    ts.append(a.value)
print(f"float32 {'with fractions ' if frac else ''}{F} x 512 x 512: n={int(dn.item())} scan stage {min(ts[1:]):.3f} ms  {mov.numel() * 4 / (min(ts[1:]) * 1e-3) / 1e9:.1f} GB/s of float32")
