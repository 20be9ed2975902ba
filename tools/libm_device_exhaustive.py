#!/usr/bin/env python3
"""The DEVICE build of csrc/libm_glibc.h (pmi_libm_eval_dev) against the host's C library on every float32 argument
(2^32 bit patterns widened to float64), exp and erf.  usage: python tools/libm_device_exhaustive.py [log2 chunk]"""
import ctypes, os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picasso_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(), "libm_glibc_host.so")
subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "picasso_amd", "csrc"),
                "-o", so, os.path.join(ROOT, "tests", "native", "libm_glibc_host.cpp")], check=True)
host = ctypes.CDLL(so)
for f in (host.ref_exp, host.ref_erf):
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
L = _lib.load()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
bad = [0, 0]
ref = np.empty(n, np.float64)
t0 = time.time()
for c in range(1 << (32 - lg)):
    bits = torch.arange(c * n, (c + 1) * n, dtype=torch.int64, device="cuda").to(torch.int32)      # wraps into the negative patterns
    x = bits.view(torch.float32).to(torch.float64)
    xh = x.cpu().numpy()
    out = torch.empty_like(x)
    for fn, rf in ((0, host.ref_exp), (1, host.ref_erf)):
        _lib.check(L.pmi_libm_eval_dev(fn, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), None), "pmi_libm_eval_dev")
        torch.cuda.synchronize()
        dev = out.cpu().numpy()
        rf(xh.ctypes.data, n, ref.ctypes.data)
        bad[fn] += int(np.count_nonzero((dev.view(np.uint64) != ref.view(np.uint64)) & ~(np.isnan(dev) & np.isnan(ref))))
print(f"device build against the host's libm on every float32 argument ({1 << 32}): exp differs on {bad[0]}, erf on {bad[1]}  ({time.time() - t0:.0f} s)")
sys.exit(1 if bad[0] or bad[1] else 0)
