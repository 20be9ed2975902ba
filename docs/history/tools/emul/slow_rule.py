"""TOOL: is the count-based `slow` flag (more than 32 iterations) still needed beside the wobble and lambda_max flags?
Escapes of the float32 loop under the device's flag set with the slow threshold at 32 / 48 / 64 / none."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from study import fuzz_spots, traces  # noqa: E402
from spec_radius import fisher_lmax  # noqa: E402


def run(box, method, eps, max_it, n, rng, style):
    spots = fuzz_spots(box, n, rng, style)
    tr, ir, tf, aux, itf = traces(spots, eps, max_it, method)
    NP = 6 if method == "sigmaxy" else 5
    tested = [0, 1, 4, 5] if NP == 6 else [0, 1]
    T = tf.shape[1]
    st = -np.diff(tf[:, :, :NP].astype(np.float64), axis=1)
    Df = np.abs(st[:, :, tested]).max(axis=2)
    kidx = np.arange(1, T)[None, :]
    valid = kidx <= itf[:, None]
    same = ir == itf
    fr = tr[np.arange(n), np.minimum(ir, T - 1)]
    ff = tf[np.arange(n), np.minimum(itf, T - 1)]
    conv = same & (ir < max_it) & np.all(np.isfinite(fr), axis=1)
    dd = np.abs(fr - ff)[:, [0, 1, 4, 5]].max(axis=1)
    with np.errstate(invalid="ignore"):
        dph = np.abs(fr[:, 2] - ff[:, 2]) / np.maximum(np.abs(fr[:, 2]), 1)
    tol = max(1e-3, eps)
    fail = (~same) | (conv & ((dd > tol) | (dph > 1e-2)))
    ulp = 2.0 ** (np.floor(np.log2(max(1.0, box / 2.0))) - 23)
    margin = max(eps * 1e-3, 4 * ulp)
    wide = np.maximum(1.0, (kidx - 1) * 0.0625)
    with np.errstate(invalid="ignore"):
        base = (valid & (np.abs(Df - eps) < margin * wide)).any(axis=1)
        base |= (valid[:, :, None] & (aux[:, :T - 1, 6:6 + NP] >= 0)).any(axis=(1, 2))
        base |= (valid[:, :, None] & (tf[:, 1:, 4:4 + (2 if NP == 6 else 1)] < 0.3)).any(axis=(1, 2))
        base |= (valid & (aux[:, :T - 1, 30] >= 16)).any(axis=1)
        w = st[:, 2:] - 2 * st[:, 1:-1] + st[:, :-2]
        vw = valid[:, 2:, None]
        th_ = np.maximum(np.abs(tf[:, 3:, :NP].astype(np.float64)), 1e-3)
        c = (w[:, 1:] * w[:, :-1] < 0) & (np.abs(w[:, 1:]) > 0.9 * np.abs(w[:, :-1])) & (np.abs(w[:, 1:]) > 1.9e-6 * th_[:, 1:]) & vw[:, 1:]
        kk3 = np.arange(4, T)[None, :, None]
        cc = c & (kk3 >= 8)
        base |= (c[:, 2:] & c[:, 1:-1] & c[:, :-2]).any(axis=(1, 2)) | (cc[:, 1:] & cc[:, :-1]).any(axis=(1, 2))
        lm = fisher_lmax(ff.astype(np.float64), box, NP)
        base |= lm > 1.9
    out = []
    for thr in (32, 48, 64, 10 ** 9):
        fl = base | (itf >= thr)
        out.append(f"slow>={thr if thr < 10**9 else 'off'}: flagged {fl.mean():.4f} escaped {int((fail & ~fl).sum())}")
    print(style, box, method, eps, max_it, "fail", int(fail.sum()), " | ".join(out), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    for style in ("real", "fuzz"):
        for box in (7, 13, 21, 5):
            for method in ("sigmaxy", "sigma"):
                for eps, max_it in ((1e-3, 100), (1e-4, 100), (1e-5, 300)):
                    run(box, method, eps, max_it, n, rng, style)
