"""TOOL: statistics of the float32-loop drift (tools/emul/mle_emul.c) on fuzz-like spots."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "_emul.so"))
P = C.c_void_p
lib.emul_traces.argtypes = [P, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, P, P, P, P, P, C.c_int]


def fuzz_spots(box, n, rng, style="fuzz"):
    c = box // 2
    idx = np.arange(box)
    x0 = c + rng.uniform(-1.5, 1.5, n); y0 = c + rng.uniform(-1.5, 1.5, n)
    if style == "fuzz":
        sx = rng.uniform(0.5, 0.3 * box + 0.5, n); sy = rng.uniform(0.5, 0.3 * box + 0.5, n)
        ph = rng.uniform(20, 9000, n); bg = rng.uniform(0.05, 60, n)
    else:   # real-like
        sx = rng.uniform(0.9, 1.4, n); sy = sx * rng.uniform(0.9, 1.1, n)
        x0 = c + rng.uniform(-0.6, 0.6, n); y0 = c + rng.uniform(-0.6, 0.6, n)
        ph = rng.uniform(2000, 8000, n); bg = rng.uniform(10, 30, n)
    gx = np.exp(-0.5 * ((idx[None, :] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
    gy = np.exp(-0.5 * ((idx[None, :] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
    lam = ph[:, None, None] * gy[:, :, None] * gx[:, None, :] + bg[:, None, None]
    spots = rng.poisson(lam).astype(np.float32)
    sub = rng.choice([0.0, 0.0, 3.0], n).astype(np.float32) if style == "fuzz" else np.zeros(n, np.float32)
    return spots - sub[:, None, None]


def traces(spots, eps, max_it, method, T=None):
    n, box, _ = spots.shape
    T = T or (max_it + 1)
    tr = np.full((n, T, 6), np.nan, np.float32); tf = np.full((n, T, 6), np.nan, np.float32)
    aux = np.full((n, T, 31), np.nan, np.float32)
    ir = np.zeros(n, np.int32); itf = np.zeros(n, np.int32)
    spots = np.ascontiguousarray(spots)
    lib.emul_traces(spots.ctypes.data, n, box, eps, max_it, 1 if method == "sigmaxy" else 0, T,
                    tr.ctypes.data, ir.ctypes.data, tf.ctypes.data, aux.ctypes.data, itf.ctypes.data, 8)
    return tr, ir, tf, aux, itf


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(5)
    for style in ("real", "fuzz"):
        for box in (7, 13, 15, 21, 3):
            for method in ("sigmaxy", "sigma"):
                for eps, max_it in ((1e-3, 100), (1e-2, 100), (1e-3, 5), (1e-4, 100)):
                    spots = fuzz_spots(box, n, rng, style)
                    tr, ir, tf, aux, itf = traces(spots, eps, max_it, method)
                    tested = [0, 1, 4, 5] if method == "sigmaxy" else [0, 1]
                    same = ir == itf
                    # final theta difference on equal counts where the reference converged
                    K = np.minimum(ir, tr.shape[1] - 1)
                    fr = tr[np.arange(n), K]; ff = tf[np.arange(n), np.minimum(itf, tf.shape[1] - 1)]
                    conv = same & (ir < max_it) & np.all(np.isfinite(fr), axis=1)
                    dd = np.abs(fr - ff)[:, [0, 1, 4, 5]].max(axis=1)
                    tol = max(1e-3, eps)
                    bad_eq = conv & (dd > tol)
                    print(f"{style} box {box} {method} eps {eps} max_it {max_it}: it differ {int((~same).sum())}, "
                          f"equal-count rows beyond {tol}: {int(bad_eq.sum())} (max {dd[conv].max() if conv.any() else 0:.2e}), "
                          f"mean it {ir.mean():.1f}", flush=True)
