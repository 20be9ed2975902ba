"""picasso.avgroi surface: the "avg" fit method (ROI sum) on the HIP backend
(picasso/avgroi.py:24-65 fit_spots, :99-164 locs_from_fits)."""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import backend, gausslq


def fit_spots(spots, progress_callback=None):
    """theta (N, 6) float32 = [0, 0, sum, sum, 1, 1] per spot."""
    theta = backend.avgroi_array(spots)
    if callable(progress_callback) and len(theta):
        progress_callback(len(theta) - 1)
    return theta


def fit_spot(spot) -> list:
    """One spot: [x, y, photons, bg, sx, sy] = [0, 0, sum, sum, 1, 1] (picasso/avgroi.py:35-41)."""
    total = float(fit_spots(np.asarray(spot, np.float32)[None])[0, 2])
    return [0, 0, total, total, 1, 1]


def fit_spots_parallel(spots, asynch: bool = False):
    """picasso/avgroi.py:66-94 without the process pool (one device call); ``asynch`` returns one finished future."""
    theta = fit_spots(spots)
    return [gausslq._DoneFuture(theta)] if asynch else theta


def fits_from_futures(futures):
    """picasso/avgroi.py:97-100."""
    return np.vstack([f.result() for f in futures])


def locs_from_fits(identifications: pd.DataFrame, theta, box: int, em) -> pd.DataFrame:
    x = theta[:, 0] + identifications["x"].to_numpy()
    y = theta[:, 1] + identifications["y"].to_numpy()
    return gausslq._table(identifications, theta, x, y, em)
