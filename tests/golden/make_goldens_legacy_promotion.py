#!/usr/bin/env python3
"""Mint the ``*_nbp.npz`` goldens: the reference's fit kernels under NUMBA's type promotion.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Usage, from the repo root:

    /opt/conda/bin/python3.9 tests/golden/make_goldens_legacy_promotion.py          # NumPy 1.26.4 (legacy promotion)
    python tests/golden/make_goldens_legacy_promotion.py --check                    # NumPy 2.x: re-mint, compare, write nothing

The real numba cannot run in this image.  What can: the reference's own
``gaussmle.py`` / ``gausslq.py``, executed as Python with the arithmetic of every ``@numba.jit``
function re-typed by numba's rules (``_nbemu``: scalar promotion, float32 ** int, array (op) scalar loop
matching, libm exp / log).  Under /opt/conda's NumPy 1.26.4 the native scalar promotion is already numba's
(``float32 (op) int -> float64``), so that interpreter is the primary one; the emulator makes the same rules
explicit, which is why the NumPy-2 interpreter reproduces the files bit for bit (``--check``: a second,
independent execution — other NumPy, other SciPy (1.15.3 vs 1.7.1), same libm).

The inputs are the spots of the committed ``gaussmle_*.npz`` / ``gausslq_*.npz`` goldens; the outputs:

    gaussmle_<set>_nbp.npz   <method>[_it3|_eps5]_{theta,crlb,loglik,iterations}, <method>_zero_division
    gausslq_<set>_nbp.npz    theta0 (gausslq._initial_parameters), theta (gausslq.fit_spot), and
                             theta_from_golden0 (the fit started from the NumPy-2 goldens' theta0)
"""
from __future__ import annotations

import argparse
import multiprocessing as mp
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _nbemu  # noqa: E402

warnings.simplefilter("ignore")
np.seterr(all="ignore")

MLE_SETS = ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "degenerate7", "poisson9",
            "poisson13", "poisson5"]
MLE_VARIANT_SETS = ("conftest_noisy", "poisson7")
LQ_SETS = ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "poisson13"]

_mods = {}


def _mod(name):
    if name not in _mods:
        _mods[name] = _nbemu.load(name)
    return _mods[name]


def _mle_one(job):
    spot, eps, max_it, method = job
    g = _mod("gaussmle")
    _nbemu.ZERO_DIVISIONS[0] = 0
    th, cr, ll, it = g.gaussmle(spot[None], eps, max_it, method)
    return th[0], cr[0], ll[0], it[0], _nbemu.ZERO_DIVISIONS[0]


def _lq_one(job):
    spot, theta0_golden = job
    q = _mod("gausslq")
    size = spot.shape[0]
    t0 = q._initial_parameters(spot, size, int(size / 2))
    th = np.asarray(q.fit_spot(spot))
    # the same fit from the committed (NumPy-2) start values: isolates the solver from the moment sums
    from scipy import optimize
    grid = np.arange(-int(size / 2), int(size / 2) + 1, dtype=np.float32)
    args = (spot, grid, size, np.empty(size, np.float32), np.empty(size, np.float32),
            np.empty((size, size), np.float32), np.empty((size, size), np.float32))
    th_g = optimize.leastsq(q._compute_residuals, theta0_golden, args=args, ftol=1e-2, xtol=1e-2)[0]
    return t0, th, np.asarray(th_g)


def mint(pool):
    out = {}
    for name in MLE_SETS:
        spots = np.load(os.path.join(HERE, f"gaussmle_{name}.npz"))["spots"]
        d = {}
        variants = [("", 1e-3, 100)]
        if name in MLE_VARIANT_SETS:
            variants += [("_it3", 1e-3, 3), ("_eps5", 1e-5, 100)]
        for method in ("sigmaxy", "sigma"):
            for tag, eps, max_it in variants:
                res = pool.map(_mle_one, [(s, eps, max_it, method) for s in spots], chunksize=2)
                d[f"{method}{tag}_theta"] = np.stack([r[0] for r in res]).astype(np.float32)
                d[f"{method}{tag}_crlb"] = np.stack([r[1] for r in res]).astype(np.float32)
                d[f"{method}{tag}_loglik"] = np.array([r[2] for r in res], np.float32)
                d[f"{method}{tag}_iterations"] = np.array([r[3] for r in res], np.int32)
                if not tag:
                    d[f"{method}_zero_division"] = np.array([r[4] for r in res], np.int32)
                print(f"   gaussmle {name:16s} {method:8s}{tag:6s} n={len(spots):4d} "
                      f"iterations mean {d[f'{method}{tag}_iterations'].mean():.2f}", flush=True)
        out[f"gaussmle_{name}_nbp"] = d
    for name in LQ_SETS:
        g = np.load(os.path.join(HERE, f"gausslq_{name}.npz"))
        res = pool.map(_lq_one, list(zip(g["spots"], g["theta0"])), chunksize=2)
        out[f"gausslq_{name}_nbp"] = {
            "theta0": np.stack([r[0] for r in res]).astype(np.float32),
            "theta": np.stack([r[1] for r in res]).astype(np.float64),
            "theta_from_golden0": np.stack([r[2] for r in res]).astype(np.float64),
        }
        print(f"   gausslq  {name:16s} n={len(res)}", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="re-mint and compare with the committed files, write nothing")
    ap.add_argument("--workers", type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args()
    import scipy
    print(f"python {sys.version.split()[0]}, numpy {np.__version__}, scipy {scipy.__version__}; "
          f"retyped: {_mod('gaussmle').__retyped__ + _mod('gausslq').__retyped__}", flush=True)
    with mp.Pool(a.workers) as pool:
        out = mint(pool)
    bad = 0
    for fname, d in out.items():
        path = os.path.join(HERE, fname + ".npz")
        if a.check:
            ref = np.load(path)
            for k, v in d.items():
                same = np.array_equal(ref[k], v, equal_nan=True)
                if not same:
                    # the CRLB goes through LAPACK (np.linalg.pinv): another NumPy build differs in the last bits, and on the
                    # (near-)singular Fisher matrices of degenerate7 in what is left of the cut-off singular values
                    lapack = False
                    if k.endswith("_crlb"):
                        with np.errstate(invalid="ignore"):
                            diff = np.abs(ref[k].astype(np.float64) - v)
                        diff[(ref[k] == v) | (np.isnan(ref[k]) & np.isnan(v))] = 0.0
                        tol = 2e-6 * np.abs(ref[k])
                        if "degenerate" in fname:
                            tol = np.maximum(tol, 1e-5 * np.nanmax(np.abs(ref[k]), axis=1, keepdims=True))
                        lapack = bool(np.all(diff <= tol))
                    print(f"   {fname}:{k} differs" + (" (CRLB only, within the LAPACK bound)" if lapack else "  <-- NOT reproduced"))
                    bad += 0 if lapack else 1
        else:
            np.savez_compressed(path, **d)
            print(f"  {fname}.npz  {os.path.getsize(path) / 1024:.1f} KiB")
    if a.check:
        print("check:", "reproduced bit for bit (CRLB columns aside where noted)" if not bad else f"{bad} arrays differ")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
