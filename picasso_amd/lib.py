"""The picasso.lib helpers the localization / undrift path depends on
(picasso/lib.py:878-920 get_from_metadata, :1786-1832 ensure_sanity, :2034-2078 minimize_shifts)."""
from __future__ import annotations

from typing import Any

import warnings

import numpy as np
import pandas as pd


def get_from_metadata(info, key: Any, default=None, *, raise_error: bool = False) -> Any:
    """Search the metadata (dict or list of dicts) from the last element to the first."""
    if isinstance(info, dict):
        info = [info]
    for d in reversed(info):
        if key in d:
            return d[key]
    if raise_error:
        raise KeyError(f"Key '{key}' not found in metadata.")
    return default


def ensure_sanity(locs: pd.DataFrame, info) -> pd.DataFrame:
    """Drop rows with inf/NaN, positions outside the image and negative
    x, y, lpx, lpy, lpz, photons, ellipticity, sx, sy (picasso/lib.py:1786-1832)."""
    locs = locs.copy()
    locs.replace([np.inf, -np.inf], np.nan, inplace=True)
    locs.dropna(axis=0, how="any", inplace=True)
    for key in ("Width", "Height", "Frames"):
        if get_from_metadata(info, key) is None:
            raise KeyError(f"Metadata is missing required key: '{key}'")
    locs = locs[locs["x"] < get_from_metadata(info, "Width")]
    locs = locs[locs["y"] < get_from_metadata(info, "Height")]
    for attr in ("x", "y", "lpx", "lpy", "lpz", "photons", "ellipticity", "sx", "sy"):
        if attr in locs.columns:
            locs = locs[locs[attr] >= 0]
    return locs


def deprecation_warning(message: str) -> None:
    warnings.warn(message, DeprecationWarning, stacklevel=3)


def minimize_shifts(shifts_x, shifts_y, shifts_z=None):
    """Least-squares consistent shifts from all pairwise shifts (picasso/lib.py:2034-2078).
    The shift between segments i < j is the sum of the step displacements D_i .. D_{j-1}, so with one
    row per pair (in the order i < j of the upper triangle) r = A D, A[row, i:j] = 1, and D = pinv(A) r.
    -> (shift_y, shift_x[, shift_z]), cumulative from segment 0."""
    n = shifts_x.shape[0]
    pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    stacked = [shifts_y, shifts_x] + ([] if shifts_z is None else [shifts_z])
    design = np.zeros((len(pairs), n - 1))
    observed = np.zeros((len(pairs), len(stacked)))
    for row, (i, j) in enumerate(pairs):
        design[row, i:j] = 1
        observed[row] = [m[i, j] for m in stacked]
    steps = np.dot(np.linalg.pinv(design), observed)
    return tuple(np.insert(np.cumsum(steps[:, d]), 0, 0) for d in range(len(stacked)))
