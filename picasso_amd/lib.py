"""The two picasso.lib helpers the localization path depends on
(picasso/lib.py:878-920 get_from_metadata, :1786-1832 ensure_sanity)."""
from __future__ import annotations

from typing import Any

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
