"""TOOL: oscillation-rule floors: false-flag rate on real-like spots, escapes on fuzz spots.
usage: python tools/emul/sweep_osc.py n style seed boxes(comma)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from study import fuzz_spots, traces  # noqa: E402


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
        base |= itf >= 32
        base |= (valid[:, :, None] & (tf[:, 1:, 4:4 + (2 if NP == 6 else 1)] < 0.3)).any(axis=(1, 2))
        base |= (valid & (aux[:, :T - 1, 30] >= 16)).any(axis=1)
        thabs = np.abs(tf[:, 1:, :NP].astype(np.float64))
        a0, a1 = st[:, :-1], st[:, 1:]
        out = {}
        # wobble: the alternating component of a parameter's step sequence, w_k = d_k - 2 d_{k-1} + d_{k-2}; an unstable
        # (or barely damped) alternating mode of the iteration map shows as w changing sign every iteration without shrinking
        w = st[:, 2:] - 2 * st[:, 1:-1] + st[:, :-2]                  # (n, T-3, NP), iteration k = index + 3
        vw = valid[:, 2:, None]
        th_ = np.maximum(np.abs(tf[:, 3:, :NP].astype(np.float64)), 1e-3)
        variants = {"none": np.zeros(n, bool)}
        kk3 = np.arange(4, T)[None, :, None]          # iteration number of w[:, 1:]
        R, fl_ = 0.9, 1.9e-6
        c = (w[:, 1:] * w[:, :-1] < 0) & (np.abs(w[:, 1:]) > R * np.abs(w[:, :-1])) & (np.abs(w[:, 1:]) > fl_ * th_[:, 1:]) & vw[:, 1:]
        cc = c & (kk3 >= 8)
        cur = (c[:, 2:] & c[:, 1:-1] & c[:, :-2]).any(axis=(1, 2)) | (cc[:, 1:] & cc[:, :-1]).any(axis=(1, 2))
        variants["current (x3 | x2 late)"] = cur
        # margin from the wobble: a tested parameter whose step lies within c x the alternating component of its own step
        # sequence of eps (the decision depends on the phase of a mode that grew out of rounding noise)
        ti = [tested.index(t) if t in tested else -1 for t in range(NP)]
        tmask = np.array([t in tested for t in range(NP)])
        alt = (w[:, 1:] * w[:, :-1] < 0) & vw[:, 1:]
        amp = np.maximum(np.abs(w[:, 1:]), np.abs(w[:, :-1]))
        stepk = np.abs(st[:, 3:, :])                   # step of the iteration of w[:, 1:]
        for cf in (1.0, 2.0, 4.0):
            am = (c & (np.abs(stepk - eps) < cf * amp) & (stepk < 2 * eps) & tmask[None, None, :]).any(axis=(1, 2))
            variants[f"cur + altmargin x{cf}"] = cur | am
        for name, osc in variants.items():
            fl = base | osc
            out[name] = (fl.mean(), int((fail & ~fl).sum()))
    print(style, box, method, eps, max_it, "fail", int(fail.sum()), "\n   " + "\n   ".join(f"{k}: fl {v[0]:.4f} esc {v[1]}" for k, v in out.items()), flush=True)


n = int(sys.argv[1])
style = sys.argv[2]
rng = np.random.default_rng(int(sys.argv[3]))
boxes = [int(b) for b in sys.argv[4].split(",")]
for box in boxes:
    for method in ("sigma", "sigmaxy"):
        for eps, max_it in ((1e-3, 100), (1e-2, 100), (1e-3, 5), (1e-4, 100)):
            run(box, method, eps, max_it, n, rng, style)
