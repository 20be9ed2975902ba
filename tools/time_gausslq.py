"""Times pmi_gausslq_dev on synthetic spots (device resident) and reports exactness vs the oracle on a sample.
usage: python tools/time_gausslq.py [N] [box] [same4|sorted]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from picasso_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
box = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(0)
c = box // 2
idx = np.arange(box) - c
base = 4096
spots = np.empty((base, box, box), np.float32)
for i in range(base):
    x0, y0 = rng.uniform(-0.8, 0.8, 2)
    sx, sy = rng.uniform(0.8, 1.4, 2)
    gx = np.exp(-0.5 * ((idx - x0) / sx) ** 2) / (np.sqrt(2 * np.pi) * sx)
    gy = np.exp(-0.5 * ((idx - y0) / sy) ** 2) / (np.sqrt(2 * np.pi) * sy)
    spots[i] = rng.poisson(rng.uniform(800, 9000) * np.outer(gy, gx) + rng.uniform(2, 40))
if len(sys.argv) > 3 and sys.argv[3] == "same4":       # four copies side by side: the groups of a wavefront stay in step
    spots = np.repeat(spots, 4, axis=0)
    spots = np.ascontiguousarray(np.tile(spots, (N // (4 * base), 1, 1)))
elif len(sys.argv) > 3 and sys.argv[3] == "sorted":     # neighbours need about the same number of evaluations
    spots = spots[np.argsort(spots.reshape(base, -1).max(axis=1))]
    spots = np.ascontiguousarray(np.repeat(spots, N // base, axis=0))
else:
    spots = np.ascontiguousarray(np.tile(spots, (N // base, 1, 1)))
N = len(spots)
L = _lib.load()
d_sp, d_th, d_nf = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
_lib.check(L.pmi_malloc(ctypes.byref(d_sp), spots.nbytes), "malloc")
_lib.check(L.pmi_malloc(ctypes.byref(d_th), N * 24), "malloc")
_lib.check(L.pmi_malloc(ctypes.byref(d_nf), N * 4), "malloc")
_lib.check(L.pmi_memcpy_h2d(d_sp, _lib.ptr(spots), spots.nbytes), "h2d")
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.pmi_gausslq_dev(d_sp, N, None, box, d_th, None, d_nf, None), "lq")
    _lib.check(L.pmi_stream_synchronize(None), "sync")
    dt = time.perf_counter() - t0
    print(f"N={N} box={box}: {dt * 1e3:.2f} ms  {N / dt / 1e6:.2f} M spots/s")
nf = np.empty(N, np.int32)
_lib.check(L.pmi_memcpy_d2h(_lib.ptr(nf), d_nf, N * 4), "d2h")
print("mean nfev", nf.mean(), "max", nf.max())
from picasso_amd import backend  # noqa: E402
print("second pass:", backend.last_lq_refit_count(), "spots;", backend.last_lq_tie_reasons())
