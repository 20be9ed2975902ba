"""Times the render and RCC kernels on device-resident data (config 3 of SURVEY.md section 8d:
the 1e6 localizations of the 10k x 512 x 512 movie rendered at oversampling 10 -> 5120^2 float32).
usage: python tools/time_render.py [N] [field] [oversampling]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from picasso_amd import _lib, backend  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
field = int(sys.argv[2]) if len(sys.argv) > 2 else 512
osamp = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
rng = np.random.default_rng(0)
sites = rng.uniform(8, field - 8, (N // 100, 2))
pick = rng.integers(0, len(sites), N)
x = (sites[pick, 0] + rng.normal(0, 0.05, N)).astype(np.float32)
y = (sites[pick, 1] + rng.normal(0, 0.05, N)).astype(np.float32)
lpx = rng.uniform(0.02, 0.08, N).astype(np.float32)
lpy = rng.uniform(0.02, 0.08, N).astype(np.float32)
L = _lib.load()


def dev(a):
    p = ctypes.c_void_p()
    _lib.check(L.pmi_malloc(ctypes.byref(p), a.nbytes), "malloc")
    _lib.check(L.pmi_memcpy_h2d(p, _lib.ptr(a), a.nbytes), "h2d")
    return p


dx, dy, dlx, dly = dev(x), dev(y), dev(lpx), dev(lpy)
for label, os_, mbw in (("gaussian os=%g" % osamp, osamp, 0.0), ("gaussian os=1 min_blur=1 (undrift segment)", 1.0, 1.0)):
    ny = nx = int(np.ceil(os_ * field))
    img = ctypes.c_void_p(); dn = ctypes.c_void_p()
    _lib.check(L.pmi_malloc(ctypes.byref(img), ny * nx * 4), "malloc")
    _lib.check(L.pmi_malloc(ctypes.byref(dn), 8), "malloc")
    for kind in ("gaussian", "hist"):
        ts = []
        for rep in range(4):
            L.pmi_stream_synchronize(None)
            t0 = time.perf_counter()
            if kind == "gaussian":
                rc = L.pmi_render_gaussian_dev(dx, dy, dlx, dly, N, os_, 0.0, 0.0, float(field), float(field), mbw, 0, img, ny, nx, dn, None)
            else:
                rc = L.pmi_render_hist_dev(dx, dy, N, os_, 0.0, 0.0, float(field), float(field), img, ny, nx, dn, None)
            _lib.check(rc, "render")
            L.pmi_stream_synchronize(None)
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        alg = N * (16 if kind == "gaussian" else 8) + ny * nx * 4
        print(f"{kind:9s} {label:44s} N={N} image {ny}x{nx}: {t * 1e3:8.3f} ms  {N / t / 1e6:8.1f} M loc/s  "
              f"algorithmic {alg / 1e6:.1f} MB -> {alg / t / 1e9:.1f} GB/s")
    L.pmi_free(img); L.pmi_free(dn)

# CPU beside it: the C restatement of the reference's sequential loop (oracle), one core
from oracle import oracle as orc  # noqa: E402
for label, os_, mbw in (("gaussian os=%g" % osamp, osamp, 0.0), ("gaussian os=1 min_blur=1", 1.0, 1.0)):
    t0 = time.perf_counter()
    orc.render(x, y, os_, [(0, 0), (field, field)], lpx, lpy, "gaussian", mbw)
    t = time.perf_counter() - t0
    print(f"cpu oracle (1 core) {label:30s}: {t * 1e3:9.1f} ms  {N / t / 1e6:7.2f} M loc/s")

# RCC: 10 segments of field x field (config 2: 10k frames, segmentation 1000)
segs = np.zeros((10, field, field))
for s in range(10):
    n, im = backend.render_arrays(x[s::10] + 0.05 * s, y[s::10], 1.0, 0, 0, field, field, lpx[s::10], lpy[s::10], 1.0)
    segs[s] = im
backend.rcc_pairs_arrays(segs, 32, 5)         # plan creation
t0 = time.perf_counter()
peak, valid, rois, crop = backend.rcc_pairs_arrays(segs, 32, 5)
t = time.perf_counter() - t0
print(f"rcc_pairs: 10 segments {field}x{field}, 45 pairs (host buffers in/out): {t * 1e3:.2f} ms")
t0 = time.perf_counter()
for i in range(9):
    for j in range(i + 1, 10):
        orc.peak_window(segs[i], segs[j], 5, 32)
t = time.perf_counter() - t0
print(f"cpu numpy restatement of the same 45 correlations + peak windows: {t * 1e3:.1f} ms")
