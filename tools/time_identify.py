#!/usr/bin/env python3
"""Time pmi_identify_dev on a resident synthetic movie for several thresholds
(main scan only / normal / every candidate exact)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from picasso_amd import _lib, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
box = int(sys.argv[2]) if len(sys.argv) > 2 else 7
L = _lib.load()
mov = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cap = 400 * F * 8
out = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(3)] + [torch.empty(cap, dtype=torch.float32, device="cuda")]
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
L.pmi_set_kernel_timing(1)
a, b = ctypes.c_float(0), ctypes.c_float(0)
modes = (("scan only (min_ng 1e12)", 1e12), ("normal (5000)", 5000.0), ("low (400)", 400.0), ("all candidates (-1e9)", -1e9))
if len(sys.argv) > 3:
    modes = (modes[int(sys.argv[3])],)
for label, ng in modes:
    ts = []
    for _ in range(4):
        rc = L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()), 0, F, 512, 512, box, ng, None, 0, F - 1,
                                *[ctypes.c_void_p(t.data_ptr()) for t in out], cap, ctypes.c_void_p(dn.data_ptr()), None)
        _lib.check(rc)
        torch.cuda.synchronize()
        L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
        ts.append(a.value)
    gb = mov.numel() * 2 / 1e9
    print(f"{label:28s} n={int(dn.item()):9d}  scan {min(ts[1:]):8.3f} ms  {gb / (min(ts[1:]) * 1e-3):8.1f} GB/s")
