"""picasso.gausslq surface: the table / precision formulas (host side).

``localization_precision`` (picasso/gausslq.py:547-589, Mortensen et al. 2010 with
the diagonal-covariance correction), ``sigma_uncertainty`` (:592-633) and
``locs_from_fits`` (:404-484) on the host, and ``fit_spots`` / ``fit_spots_parallel``
(:247-401): the per-spot scipy.optimize.leastsq of the reference (MINPACK lmdif,
:206-244) runs as one HIP kernel over all spots (csrc/gausslq.hip).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import backend


def fit_spots(spots, progress_callback=None):
    """theta (N, 6) float32 = x, y, photons, bg, sx, sy; x, y relative to the box
    centre (picasso/gausslq.py:247-268).  One kernel launch fits every spot, so a
    callable progress_callback is called once, when the fit is done."""
    theta = backend.gausslq_arrays(spots)
    if callable(progress_callback) and len(theta):
        progress_callback(len(theta) - 1)     # the reference reports the index of the last fitted spot
    return theta


def fit_spot(spot):
    """One spot (picasso/gausslq.py:206-244); float64 like leastsq's result."""
    return backend.gausslq_arrays(np.asarray(spot, np.float32)[None])[0].astype(np.float64)


class _DoneFuture:
    """concurrent.futures-shaped result holder for fit_spots_parallel(asynch=True)."""

    def __init__(self, value):
        self._value = value

    def done(self):
        return True

    def result(self, timeout=None):
        return self._value


def fit_spots_parallel(spots, asynch: bool = False):
    """picasso/gausslq.py:271-314.  The reference splits the spots over a process
    pool; here the GPU fits them all in one call.  With asynch=True a list holding
    one finished future is returned (the shape fits_from_futures expects)."""
    theta = fit_spots(spots)
    if asynch:
        return [_DoneFuture(theta)]
    return theta


def fits_from_futures(futures):
    """picasso/gausslq.py:317-333."""
    return np.vstack([f.result() for f in futures])


def localization_precision(photons, s, s_orth, bg, em: bool):
    s2 = s**2
    sa2 = s2 + 1 / 12
    sa = sa2**0.5
    sa_orth2 = s_orth**2 + 1 / 12
    sa_orth = sa_orth2**0.5
    v = sa2 * (16 / 9 + (8 * np.pi * sa * sa_orth * bg) / photons) / photons
    if em:
        v *= 2
    with np.errstate(invalid="ignore"):
        return np.sqrt(v)


def sigma_uncertainty(sigma, sigma_orth, photons, bg):
    sa2 = sigma**2 + 1 / 12
    sa4 = sa2**2
    sa = sa2**0.5
    sa2_orth = sigma_orth**2 + 1 / 12
    sa_orth = sa2_orth**0.5
    var_sa2 = sa4 / photons * (512 / 81 + (64 * np.pi * sa * sa_orth * bg) / (3 * photons))
    var_sigma = var_sa2 / (4 * sigma**2)
    return np.sqrt(var_sigma)


def _table(identifications: pd.DataFrame, theta, x, y, em: bool) -> pd.DataFrame:
    lpx = localization_precision(theta[:, 2], theta[:, 4], theta[:, 5], theta[:, 3], em=em)
    lpy = localization_precision(theta[:, 2], theta[:, 5], theta[:, 4], theta[:, 3], em=em)
    a = np.maximum(theta[:, 4], theta[:, 5])
    b = np.minimum(theta[:, 4], theta[:, 5])
    cols = {
        "frame": identifications["frame"].to_numpy().astype(np.uint32),
        "x": np.asarray(x).astype(np.float32),
        "y": np.asarray(y).astype(np.float32),
        "photons": theta[:, 2].astype(np.float32),
        "sx": theta[:, 4].astype(np.float32),
        "sy": theta[:, 5].astype(np.float32),
        "bg": theta[:, 3].astype(np.float32),
        "lpx": lpx.astype(np.float32),
        "lpy": lpy.astype(np.float32),
        "ellipticity": ((a - b) / a).astype(np.float32),
        "net_gradient": identifications["net_gradient"].to_numpy().astype(np.float32),
    }
    if "n_id" in identifications.columns:
        cols["n_id"] = identifications["n_id"].to_numpy().astype(np.uint32)
        locs = pd.DataFrame(cols)
        locs.sort_values(by="n_id", kind="quicksort", inplace=True)
    else:
        locs = pd.DataFrame(cols)
        locs.sort_values(by="frame", kind="quicksort", inplace=True)
    return locs


def locs_from_fits(identifications: pd.DataFrame, theta, box: int, em: bool) -> pd.DataFrame:
    """11 columns (+ n_id); x = theta_x + id.x — no box offset, the LQ theta is centre-relative."""
    x = theta[:, 0] + identifications["x"].to_numpy()
    y = theta[:, 1] + identifications["y"].to_numpy()
    return _table(identifications, theta, x, y, em)
