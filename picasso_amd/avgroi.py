"""picasso.avgroi surface: the "avg" fit method (ROI sum) on the HIP backend
(picasso/avgroi.py:24-65 fit_spots, :99-164 locs_from_fits)."""
from __future__ import annotations

import pandas as pd

from . import backend, gausslq


def fit_spots(spots, progress_callback=None):
    """theta (N, 6) float32 = [0, 0, sum, sum, 1, 1] per spot."""
    theta = backend.avgroi_array(spots)
    if callable(progress_callback) and len(theta):
        progress_callback(len(theta) - 1)
    return theta


def locs_from_fits(identifications: pd.DataFrame, theta, box: int, em) -> pd.DataFrame:
    x = theta[:, 0] + identifications["x"].to_numpy()
    y = theta[:, 1] + identifications["y"].to_numpy()
    return gausslq._table(identifications, theta, x, y, em)
