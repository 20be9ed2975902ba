"""picasso.gaussmle surface on the HIP backend.

Same names, arguments, defaults, return types and error behaviour as the
reference module (picasso/gaussmle.py:409-530, 957-1074); the per-spot Newton
fit runs in libpicasso_hip.so (csrc/gaussmle.hip).  Host-side table assembly
(`locs_from_fits`) is numpy/pandas like the reference's.
"""
from __future__ import annotations

import threading
from typing import Callable, Literal

import numpy as np
import pandas as pd

from . import backend

_BATCH = 1 << 18   # spots per device call: the granularity of progress / abort polling


def _check_method(method):
    if method not in ("sigma", "sigmaxy"):
        raise ValueError("Method not available.")          # gaussmle.py:465


def gaussmle(spots, eps: float, max_it: int, method: Literal["sigma", "sigmaxy"] = "sigmaxy",
             progress_callback: Callable[[int], None] | Literal["console"] | None = None):
    """Fit Gaussians by MLE to the extracted spots (picasso/gaussmle.py:409-475).

    Returns thetas (N,6) float32, CRLBs (N,6) float32, likelihoods (N,) float32,
    iterations (N,) int32.  ``progress_callback`` receives the index of the last
    fitted spot of each device batch (the reference calls it per spot; callers
    only need a monotonically increasing count).
    """
    _check_method(method)
    spots = np.ascontiguousarray(spots, dtype=np.float32)
    N = len(spots)
    thetas = np.zeros((N, 6), dtype=np.float32)
    CRLBs = np.inf * np.ones((N, 6), dtype=np.float32)
    likelihoods = np.zeros(N, dtype=np.float32)
    iterations = np.zeros(N, dtype=np.int32)
    bar = None
    if progress_callback == "console":
        from tqdm import tqdm
        bar = tqdm(total=N, desc="Fitting...", unit="spot")
    for i0 in range(0, N, _BATCH):
        i1 = min(N, i0 + _BATCH)
        th, cr, ll, it = backend.gaussmle_arrays(spots[i0:i1], eps, max_it, method)
        thetas[i0:i1], CRLBs[i0:i1], likelihoods[i0:i1], iterations[i0:i1] = th, cr, ll, it
        if bar is not None:
            bar.update(i1 - i0)
        elif callable(progress_callback):
            progress_callback(i1 - 1)
    if bar is not None:
        bar.close()
    return thetas, CRLBs, likelihoods, iterations


def gaussmle_async(spots, eps: float, max_it: int, method: Literal["sigma", "sigmaxy"] = "sigmaxy"):
    """Asynchronous form (picasso/gaussmle.py:478-530): returns ``[current]`` — a
    one-element list counting fitted spots, N when finished — plus the output
    arrays being filled by a background thread (the ctypes call releases the GIL)."""
    _check_method(method)
    spots = np.ascontiguousarray(spots, dtype=np.float32)
    N = len(spots)
    thetas = np.zeros((N, 6), dtype=np.float32)
    CRLBs = np.inf * np.ones((N, 6), dtype=np.float32)
    likelihoods = np.zeros(N, dtype=np.float32)
    iterations = np.zeros(N, dtype=np.int32)
    current = [0]
    failure = []

    def work():
        try:
            for i0 in range(0, N, _BATCH):
                i1 = min(N, i0 + _BATCH)
                th, cr, ll, it = backend.gaussmle_arrays(spots[i0:i1], eps, max_it, method)
                thetas[i0:i1], CRLBs[i0:i1], likelihoods[i0:i1], iterations[i0:i1] = th, cr, ll, it
                current[0] = i1
        except BaseException as exc:      # surfaced by wait_async(); never swallowed
            failure.append(exc)
            current[0] = N

    t = threading.Thread(target=work, name="picasso_amd-gaussmle", daemon=True)
    t.start()
    current_failure[id(current)] = (t, failure)
    return current, thetas, CRLBs, likelihoods, iterations


current_failure: dict = {}


def wait_async(current):
    """Join the worker behind a ``gaussmle_async`` counter and re-raise its error."""
    t, failure = current_failure.pop(id(current), (None, []))
    if t is not None:
        t.join()
    if failure:
        raise failure[0]


def locs_from_fits(identifications: pd.DataFrame, theta, CRLBs, log_likelihoods, iterations, box: int) -> pd.DataFrame:
    """Fit results -> localization table (picasso/gaussmle.py:957-1037): 17 columns (+ n_id).
    Positions are theta + identification - box // 2 (theta is in box-origin coordinates), precisions
    and parameter uncertainties are the square roots of the CRLB columns, everything float32 except
    frame / iterations / n_id (uint32); sorted by n_id if present, else by frame."""
    f32, u32 = np.float32, np.uint32
    half = int(box / 2)
    with np.errstate(invalid="ignore"):
        root = np.sqrt(CRLBs)                        # columns: x, y, photons, bg, sx, sy
        wide, narrow = np.maximum(theta[:, 4], theta[:, 5]), np.minimum(theta[:, 4], theta[:, 5])
        ellipticity = (wide - narrow) / wide
    columns = [
        ("frame", identifications["frame"].to_numpy(dtype=u32)),
        ("x", (theta[:, 0] + identifications["x"] - half).astype(f32)),       # float32 + int64 -> float64, then cast
        ("y", (theta[:, 1] + identifications["y"] - half).astype(f32)),
        ("photons", theta[:, 2].astype(f32)), ("sx", theta[:, 4].astype(f32)), ("sy", theta[:, 5].astype(f32)),
        ("bg", theta[:, 3].astype(f32)), ("lpx", root[:, 0].astype(f32)), ("lpy", root[:, 1].astype(f32)),
        ("ellipticity", ellipticity.astype(f32)),
        ("net_gradient", identifications["net_gradient"].astype(f32)),
        ("log_likelihood", log_likelihoods.astype(f32)), ("iterations", iterations.astype(u32)),
        ("photons_unc", root[:, 2].astype(f32)), ("bg_unc", root[:, 3].astype(f32)),
        ("sx_unc", root[:, 4].astype(f32)), ("sy_unc", root[:, 5].astype(f32)),
    ]
    locs = pd.DataFrame(dict(columns))
    key = "frame"
    if "n_id" in identifications.columns:
        locs["n_id"] = identifications.n_id.astype(u32)
        key = "n_id"
    locs.sort_values(by=[key], kind="quicksort", inplace=True)
    return locs


def sigma_uncertainty(sigma, sigma_orth, photons, bg):
    """Standard error of the fitted sigma (picasso/gaussmle.py:1040-1074,
    Rieger & Stallinga 2014 approximation)."""
    sa2 = sigma**2 + 1 / 12
    tau = (2 * np.pi * sa2 * bg) / (photons)
    delta_sigma_sq = (sigma**2 / (4 * photons)) * (1 + 8 * tau + np.sqrt((8 * tau) / (1 + 2 * tau)))
    return np.sqrt(delta_sigma_sq)
