"""Times pmi_gaussmle_dev (spots resident) for a box size.  usage: python tools/time_gaussmle.py [N] [box] [method]"""
import ctypes
import sys
import time
from math import erf, sqrt

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from picasso_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
box = int(sys.argv[2]) if len(sys.argv) > 2 else 13
method = sys.argv[3] if len(sys.argv) > 3 else "sigmaxy"
rng = np.random.default_rng(0)
c = box // 2
idx = np.arange(box)
base = 2048
spots = np.empty((base, box, box), np.float32)
for i in range(base):
    x0, y0 = c + rng.uniform(-0.5, 0.5, 2)
    sx, sy = rng.uniform(1.0, 0.2 * box, 2)
    ex = np.array([0.5 * (erf((k - x0 + .5) / (sqrt(2) * sx)) - erf((k - x0 - .5) / (sqrt(2) * sx))) for k in idx])
    ey = np.array([0.5 * (erf((k - y0 + .5) / (sqrt(2) * sy)) - erf((k - y0 - .5) / (sqrt(2) * sy))) for k in idx])
    spots[i] = rng.poisson(rng.uniform(3000, 9000) * np.outer(ey, ex) + rng.uniform(5, 25))
spots = np.ascontiguousarray(np.tile(spots, (N // base, 1, 1)))
N = len(spots)
L = _lib.load()


def dmalloc(nbytes):
    p = ctypes.c_void_p()
    _lib.check(L.pmi_malloc(ctypes.byref(p), nbytes), "malloc")
    return p


d_sp = dmalloc(spots.nbytes)
_lib.check(L.pmi_memcpy_h2d(d_sp, _lib.ptr(spots), spots.nbytes), "h2d")
d_th, d_cr, d_ll, d_it = dmalloc(N * 24), dmalloc(N * 24), dmalloc(N * 4), dmalloc(N * 4)
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.pmi_gaussmle_dev(d_sp, N, None, box, 1e-3, 100, _lib.MLE_METHODS[method], d_th, d_cr, d_ll, d_it, None), "mle")
    _lib.check(L.pmi_stream_synchronize(None), "sync")
    dt = time.perf_counter() - t0
    print(f"N={N} box={box} {method}: {dt * 1e3:.2f} ms  {N / dt / 1e6:.2f} M spots/s")
it = np.empty(N, np.int32)
_lib.check(L.pmi_memcpy_d2h(_lib.ptr(it), d_it, N * 4), "d2h")
print("mean iterations", it.mean(), "max", it.max())
