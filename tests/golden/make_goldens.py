#!/usr/bin/env python3
"""Mint golden vectors from the reference's own source (build container only).

TEST INFRASTRUCTURE.  Usage (from the repo root, in the container that has
/root/reference mounted):

    MPLBACKEND=Agg python tests/golden/make_goldens.py

Writes ``tests/golden/*.npz`` — inputs and the reference's outputs for
identify / get_spots / gaussmle / locs_from_fits / gausslq.  The reference
source is executed in place through ``_refshim`` (NumPy semantics, see that
module's docstring); no reference text is copied.  ``testdata_movie.npz`` holds
the reference's bundled test movie (tests/data/testdata.raw, a data fixture).
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

warnings.simplefilter("ignore")
np.seterr(all="ignore")

ref = _refshim.load_reference()
localize, gaussmle, gausslq = ref["localize"], ref["gaussmle"], ref["gausslq"]
CAM = {"Baseline": 0, "Sensitivity": 1, "Gain": 1}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def ids_arrays(ids: pd.DataFrame, prefix=""):
    return {
        prefix + "frame": ids["frame"].to_numpy(np.int64),
        prefix + "y": ids["y"].to_numpy(np.int64),
        prefix + "x": ids["x"].to_numpy(np.int64),
        prefix + "ng": ids["net_gradient"].to_numpy(np.float32),
    }


# ---------------------------------------------------------------------------
# spot generators (the conftest generators are re-created from their seeds;
# tests/conftest.py:68-89,121-188 of the reference)
# ---------------------------------------------------------------------------
def make_gaussian_spot(box, x0, y0, sx, sy, photons, bg):
    half = box // 2
    grid = np.arange(-half, half + 1, dtype=np.float64)
    gx = np.exp(-0.5 * ((grid - x0) / sx) ** 2) / (sx * np.sqrt(2 * np.pi))
    gy = np.exp(-0.5 * ((grid - y0) / sy) ** 2) / (sy * np.sqrt(2 * np.pi))
    return (photons * np.outer(gy, gx) + bg).astype(np.float32)


def conftest_synthetic_spots():
    box, n = 7, 64
    rng = np.random.default_rng(42)
    gt = dict(
        x=rng.uniform(-0.5, 0.5, n), y=rng.uniform(-0.5, 0.5, n),
        sx=rng.uniform(0.9, 1.4, n), sy=rng.uniform(0.9, 1.4, n),
        photons=rng.uniform(2000.0, 8000.0, n), bg=rng.uniform(5.0, 30.0, n),
    )
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        spots[i] = make_gaussian_spot(box, gt["x"][i], gt["y"][i], gt["sx"][i],
                                      gt["sy"][i], gt["photons"][i], gt["bg"][i])
    return spots, gt


def conftest_synthetic_spots_noisy():
    box, n = 7, 32
    rng = np.random.default_rng(123)
    gt = dict(
        x=rng.uniform(-0.5, 0.5, n), y=rng.uniform(-0.5, 0.5, n),
        sx=rng.uniform(0.9, 1.4, n), sy=rng.uniform(0.9, 1.4, n),
        photons=rng.uniform(5000.0, 12000.0, n), bg=rng.uniform(5.0, 20.0, n),
    )
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        clean = make_gaussian_spot(box, gt["x"][i], gt["y"][i], gt["sx"][i],
                                   gt["sy"][i], gt["photons"][i], gt["bg"][i])
        spots[i] = rng.poisson(np.maximum(clean, 0.0)).astype(np.float32)
    return spots, gt


def erf_spot(box, x0, y0, sx, sy, photons, bg):
    """Pixel-integrated Gaussian, centre in box-origin coordinates."""
    from math import erf, sqrt
    idx = np.arange(box)
    ex = np.array([0.5 * (erf((i - x0 + 0.5) / (sqrt(2) * sx))
                          - erf((i - x0 - 0.5) / (sqrt(2) * sx))) for i in idx])
    ey = np.array([0.5 * (erf((j - y0 + 0.5) / (sqrt(2) * sy))
                          - erf((j - y0 - 0.5) / (sqrt(2) * sy))) for j in idx])
    return photons * np.outer(ey, ex) + bg


def poisson_batch(box, n, seed, photons=(300.0, 8000.0), bg=(1.0, 40.0),
                  sig=(0.8, 1.6), off=0.9):
    rng = np.random.default_rng(seed)
    c = box // 2
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        lam = erf_spot(box, c + rng.uniform(-off, off), c + rng.uniform(-off, off),
                       rng.uniform(*sig), rng.uniform(*sig),
                       rng.uniform(*photons), rng.uniform(*bg))
        spots[i] = rng.poisson(lam).astype(np.float32)
    return spots


def degenerate_batch(box=7):
    rng = np.random.default_rng(7)
    c = box // 2
    out = []
    out.append(np.full((box, box), 10.0, np.float32))            # flat
    out.append(np.zeros((box, box), np.float32))                 # all zero
    s = np.zeros((box, box), np.float32); s[c, c] = 500.0        # single hot px
    out.append(s)
    s = np.full((box, box), 5.0, np.float32); s[0, 0] = 900.0    # corner hot px
    out.append(s)
    s = rng.poisson(3.0, (box, box)).astype(np.float32)          # pure noise
    out.append(s)
    s = rng.poisson(erf_spot(box, c, c, 1.1, 1.1, 4000, 20)).astype(np.float32) - 30.0
    out.append(s.astype(np.float32))                             # negatives after baseline
    s = rng.poisson(erf_spot(box, 0.2, c, 1.2, 1.0, 3000, 10)).astype(np.float32)
    out.append(s)                                                # emitter at the left edge
    s = rng.poisson(erf_spot(box, c, c, 3.5, 3.5, 6000, 10)).astype(np.float32)
    out.append(s)                                                # very wide
    s = rng.poisson(erf_spot(box, c + 0.3, c - 0.2, 0.35, 0.4, 5000, 2)).astype(np.float32)
    out.append(s)                                                # very narrow
    s = rng.poisson(erf_spot(box, c, c, 1.0, 1.0, 30, 1)).astype(np.float32)
    out.append(s)                                                # very dim
    two = erf_spot(box, 1.5, 1.5, 1.0, 1.0, 2500, 8) + erf_spot(box, 5.0, 4.5, 1.0, 1.0, 2500, 0)
    out.append(rng.poisson(two).astype(np.float32))              # two emitters
    s = np.tile(np.arange(box, dtype=np.float32) * 10.0, (box, 1))
    out.append(s)                                                # linear ramp
    return np.stack(out)


def run_mle(spots, eps, max_it, method):
    th, cr, ll, it = gaussmle.gaussmle(spots, eps, max_it, method)
    return dict(theta=th, crlb=cr, loglik=ll, iterations=it)


def main():
    movie = _refshim.load_test_movie()
    print("movie", movie.shape, movie.dtype)
    save("testdata_movie", movie=movie)

    # ------------------------------------------------------------- identify
    print("identify goldens")
    cases = {}
    params = {
        "a": dict(min_ng=5000, box=7),
        "b": dict(min_ng=2000, box=5),
        "c": dict(min_ng=3000, box=9),
        "d": dict(min_ng=400, box=7),
        "e": dict(min_ng=5000, box=7, roi=((2, 1), (31, 32))),
        "f": dict(min_ng=5000, box=7, frame_bounds=(10, 50)),
        "g": dict(min_ng=-1e9, box=7),   # every local maximum, incl. noise maxima
        "h": dict(min_ng=1000, box=3),
    }
    for key, p in params.items():
        ids = localize.identify(movie, p["min_ng"], p["box"], roi=p.get("roi"),
                                frame_bounds=p.get("frame_bounds"),
                                threaded=False, return_info=False)
        print(f"   case {key}: {p} -> {len(ids)} ids")
        cases.update(ids_arrays(ids, key + "_"))
        cases[key + "_min_ng"] = np.float64(p["min_ng"])
        cases[key + "_box"] = np.int64(p["box"])
        cases[key + "_roi"] = np.array(p["roi"], np.int64).ravel() if "roi" in p else np.zeros(0, np.int64)
        fb = p.get("frame_bounds")
        cases[key + "_frame_bounds"] = np.array(fb, np.int64) if fb else np.zeros(0, np.int64)
    save("identify_testdata", **cases)

    # a small adversarial movie: plateaus/ties, maxima on the y==h / x==h band
    # (negative-index wrap of picasso/localize.py:179-180), non-square frames
    rng = np.random.default_rng(20261001)
    adv = rng.integers(0, 6, size=(6, 40, 52)).astype(np.uint16) * 50  # many ties
    for f in range(6):
        for _ in range(10):
            y = int(rng.integers(3, 40 - 4)); x = int(rng.integers(3, 52 - 4))
            adv[f, y - 1:y + 2, x - 1:x + 2] += np.uint16(rng.integers(200, 1500))
            adv[f, y, x] += np.uint16(rng.integers(1, 900))
        # forced candidates on the wrap band and the far band
        adv[f, 3, 3] = 6000; adv[f, 3, 25] = 5000; adv[f, 20, 3] = 5500
        adv[f, 40 - 5, 52 - 5] = 6100; adv[f, 40 - 4, 10] = 7000  # second one is outside the scan range
        adv[f, 39, :] += 300  # last row feeds the wrapped read of row -1
        adv[f, :, 51] += 200  # last column feeds the wrapped read of col -1
    adv_cases = {"movie": adv}
    for key, p in {"a": dict(min_ng=300, box=7), "b": dict(min_ng=100, box=5),
                   "c": dict(min_ng=-1e9, box=7), "d": dict(min_ng=200, box=9,
                                                            roi=((2, 5), (38, 50)))}.items():
        ids = localize.identify(adv, p["min_ng"], p["box"], roi=p.get("roi"),
                                threaded=False, return_info=False)
        print(f"   adversarial {key}: {p} -> {len(ids)} ids")
        adv_cases.update(ids_arrays(ids, key + "_"))
        adv_cases[key + "_min_ng"] = np.float64(p["min_ng"])
        adv_cases[key + "_box"] = np.int64(p["box"])
        adv_cases[key + "_roi"] = np.array(p["roi"], np.int64).ravel() if "roi" in p else np.zeros(0, np.int64)
    save("identify_adversarial", **adv_cases)

    # ------------------------------------------------------------ get_spots
    print("get_spots goldens")
    ids_a = localize.identify(movie, 5000, 7, threaded=False, return_info=False)
    gs = {}
    gs.update(ids_arrays(ids_a))
    for key, cam in {"unit": CAM,
                     "emccd": {"Baseline": 100, "Sensitivity": 0.45, "Gain": 3},
                     "scmos": {"Baseline": 99.5, "Sensitivity": 0.23, "Gain": 1}}.items():
        gs["spots_" + key] = localize.get_spots(movie, ids_a, 7, cam)
        gs["cam_" + key] = np.array([cam["Baseline"], cam["Sensitivity"], cam["Gain"]], np.float64)
    ids_c = localize.identify(movie, 3000, 9, threaded=False, return_info=False)
    gs.update(ids_arrays(ids_c, "box9_"))
    gs["box9_spots"] = localize.get_spots(movie, ids_c, 9, CAM)
    save("get_spots_testdata", **gs)

    # ------------------------------------------------------------- gaussmle
    print("gaussmle goldens (slow: pure-Python execution of the reference)")
    real_spots = gs["spots_unit"]
    datasets = {
        "conftest_clean": conftest_synthetic_spots()[0],
        "conftest_noisy": conftest_synthetic_spots_noisy()[0],
        "testdata_real": real_spots,
        "poisson7": poisson_batch(7, 192, 1001),
        "degenerate7": degenerate_batch(7),
        "poisson9": poisson_batch(9, 48, 1002, sig=(0.9, 2.0)),
        "poisson13": poisson_batch(13, 64, 1003, photons=(1500.0, 12000.0),
                                   sig=(1.0, 2.6), off=1.5),
        "poisson5": poisson_batch(5, 32, 1004, sig=(0.7, 1.1), off=0.5),
    }
    for name, spots in datasets.items():
        out = {"spots": spots}
        for method in ("sigmaxy", "sigma"):
            r = run_mle(spots, 1e-3, 100, method)
            out.update({f"{method}_{k}": v for k, v in r.items()})
            print(f"   {name:16s} {method:8s} n={len(spots):4d} "
                  f"iters mean {r['iterations'].mean():.2f} max {r['iterations'].max()}")
        if name in ("conftest_noisy", "poisson7"):
            for method in ("sigmaxy", "sigma"):
                r = run_mle(spots, 1e-3, 3, method)       # iteration-limited
                out.update({f"{method}_it3_{k}": v for k, v in r.items()})
                r = run_mle(spots, 1e-5, 100, method)     # tighter eps
                out.update({f"{method}_eps5_{k}": v for k, v in r.items()})
        save("gaussmle_" + name, **out)

    # ------------------------------------------------------- locs_from_fits
    print("locs_from_fits goldens")
    r = run_mle(real_spots, 1e-3, 100, "sigmaxy")
    locs = gaussmle.locs_from_fits(ids_a, r["theta"], r["crlb"], r["loglik"],
                                   r["iterations"], 7)
    out = {c: locs[c].to_numpy() for c in locs.columns}
    out["columns"] = np.array(list(locs.columns))
    out["dtypes"] = np.array([str(locs[c].dtype) for c in locs.columns])
    save("locs_from_fits_mle", **out)

    # -------------------------------------------------------------- gausslq
    print("gausslq goldens (scipy MINPACK lmdif through the reference)")
    for name in ("conftest_clean", "conftest_noisy", "testdata_real", "poisson7",
                 "poisson13"):
        spots = datasets[name]
        theta = gausslq.fit_spots(spots)
        theta0 = np.stack([gausslq._initial_parameters(s, s.shape[0], s.shape[0] // 2)
                           for s in spots])
        save("gausslq_" + name, spots=spots, theta=theta, theta0=theta0)
    theta = gausslq.fit_spots(real_spots)
    for em in (False, True):
        locs = gausslq.locs_from_fits(ids_a, theta, 7, em)
        out = {c: locs[c].to_numpy() for c in locs.columns}
        out["columns"] = np.array(list(locs.columns))
        out["dtypes"] = np.array([str(locs[c].dtype) for c in locs.columns])
        save("locs_from_fits_lq_em%d" % int(em), **out)
    # ----------------------------------------------------------------- zfit
    print("zfit goldens (scipy bounded Brent through the reference)")
    zfit, avgroi = ref["zfit"], ref["avgroi"]
    CALIB = {   # reference tests/conftest.py:207-229 (CALIB_3D)
        "X Coefficients": [-1.6680708772714857e-18, 2.4038209829154137e-15, 2.1771067332017187e-12,
                           -3.0324788231238476e-09, 3.5433326085494675e-06, 0.0023039289366630425, 1.2026032603707493],
        "Y Coefficients": [-1.7708672355491796e-18, 9.808249540501714e-16, 2.10653248543535e-12,
                           2.228026137415219e-11, 3.628007433361433e-06, -0.001646865504353452, 1.2257249554338714],
        "Step size in nm": 5.0, "Number of frames": 201, "Magnification factor": 0.79,
    }
    rng = np.random.default_rng(2026)
    n = 400
    ztrue = rng.uniform(-450, 450, n)
    cxv, cyv = np.array(CALIB["X Coefficients"]), np.array(CALIB["Y Coefficients"])
    wx = np.polyval(cxv, ztrue) * rng.normal(1.0, 0.03, n)
    wy = np.polyval(cyv, ztrue) * rng.normal(1.0, 0.03, n)
    wx[:6] = [0.3, 3.5, 1.0, 1e-3, 2.9, 0.0]      # off-curve / degenerate widths
    wy[:6] = [3.5, 0.3, 1.0, 1e-3, 2.9, 1.2]
    zl = pd.DataFrame({
        "frame": np.sort(rng.integers(0, 50, n)).astype(np.uint32),
        "x": rng.uniform(5, 27, n).astype(np.float32), "y": rng.uniform(5, 27, n).astype(np.float32),
        "photons": rng.uniform(800, 9000, n).astype(np.float32),
        "sx": wx.astype(np.float32), "sy": wy.astype(np.float32),
        "bg": rng.uniform(2, 40, n).astype(np.float32),
        "lpx": rng.uniform(0.005, 0.05, n).astype(np.float32), "lpy": rng.uniform(0.005, 0.05, n).astype(np.float32),
        "sx_unc": rng.uniform(0.005, 0.05, n).astype(np.float32), "sy_unc": rng.uniform(0.005, 0.05, n).astype(np.float32),
    })
    zinfo = [{"Width": 32, "Height": 32, "Frames": 50, "Pixelsize": 130}]
    out = {c: zl[c].to_numpy() for c in zl.columns}
    out["cx"], out["cy"], out["magnification"], out["pixelsize"] = cxv, cyv, np.float64(0.79), np.float64(130.0)
    for method in ("gausslq", "gaussmle"):
        r = zfit._fit_z(zl, zinfo, CALIB, 0.79, 130, fitting_method=method, filter=0)
        out[method + "_index"] = r.index.to_numpy()      # rows surviving ensure_sanity
        for c in ("z", "d_zcalib", "lpz"):
            out[f"{method}_{c}"] = r[c].to_numpy()
        r2 = zfit._fit_z(zl, zinfo, CALIB, 0.79, 130, fitting_method=method, filter=2)
        out[method + "_index_filter2"] = r2.index.to_numpy()
    save("zfit_calib3d", **out)

    # ------------------------------------------------------------------ avg
    theta = avgroi.fit_spots(real_spots)
    locs = avgroi.locs_from_fits(ids_a, theta, 7, False)
    o2 = {c: locs[c].to_numpy() for c in locs.columns}
    o2["theta"] = theta
    o2["columns"] = np.array(list(locs.columns))
    save("avg_testdata", **o2)
    print("done")


if __name__ == "__main__":
    main()
