"""The RCC drift-correction entry points of picasso.postprocess (picasso/postprocess.py:2824-2961
``n_segments``, ``segment``, ``undrift``; :3157-3218 ``_apply_drift`` / ``apply_drift``) on top of
the GPU render (csrc/render.hip) and cross-correlation (csrc/xcorr.hip).
"""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import interpolate

from . import imageprocess, lib, render


def n_segments(info, segmentation: int) -> int:
    n_frames = lib.get_from_metadata(info, "Frames")
    return int(np.round(n_frames / segmentation))


def segment(locs: pd.DataFrame, info, segmentation: int, kwargs: dict = {}, callback=None):
    """Split the localizations into temporal segments and render each one (postprocess.py:2846-2897).
    -> bounds (uint32 frame bounds), segments (n_seg, Y, X) float64."""
    Y = info[0]["Height"]
    X = info[0]["Width"]
    n_frames = info[0]["Frames"]
    n_seg = n_segments(info, segmentation)
    bounds = np.linspace(0, n_frames - 1, n_seg + 1, dtype=np.uint32)
    segments = np.zeros((n_seg, Y, X))
    if callback is not None:
        callback(0)
    for i in range(n_seg):
        segment_locs = locs[(locs["frame"] >= bounds[i]) & (locs["frame"] < bounds[i + 1])]
        _, segments[i] = render.render(segment_locs, info, **kwargs)
        if callback is not None:
            callback(i + 1)
    return bounds, segments


def _apply_drift(locs: pd.DataFrame, drift: pd.DataFrame) -> pd.DataFrame:
    frames = locs["frame"]
    locs["x"] -= drift["x"].iloc[frames].to_numpy()
    locs["y"] -= drift["y"].iloc[frames].to_numpy()
    if "z" in drift.columns and "z" in locs.columns:
        locs["z"] -= drift["z"].iloc[frames].to_numpy()
    return locs


def apply_drift(locs: pd.DataFrame, info, *, drift):
    assert isinstance(drift, (pd.DataFrame, np.ndarray)), "Drift must be a DataFrame or numpy array"
    n_frames = lib.get_from_metadata(info, "Frames", raise_error=True)
    if isinstance(drift, pd.DataFrame):
        required_columns = {"x", "y"}
        if not required_columns.issubset(drift.columns):
            raise ValueError(f"Drift DataFrame must contain columns {required_columns}")
    elif isinstance(drift, np.ndarray):
        if not (drift.shape[1] in [2, 3] and drift.shape[0] == n_frames):
            raise ValueError("Drift array must have shape (n_frames, 2) for x and y drift, "
                             "or (n_frames, 3) for x, y, and z drift.")
        drift = pd.DataFrame(drift, columns=["x", "y"] + (["z"] if drift.shape[1] == 3 else []))
    return _apply_drift(locs, drift)


def undrift(locs: pd.DataFrame, info, segmentation: int, display: bool = True, segmentation_callback=None,
            rcc_callback=None):
    """RCC drift correction (postprocess.py:2900-2961) -> (drift DataFrame, undrifted locs).
    ``display`` (a matplotlib plot in the reference) is ignored."""
    import warnings
    locs = locs.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)       # render()'s oversampling notice, as in the reference call
        bounds, segments = segment(locs, info, segmentation, {"blur_method": "gaussian", "min_blur_width": 1},
                                   segmentation_callback)
    shift_y, shift_x = imageprocess.rcc(segments, 32, rcc_callback)
    t = (bounds[1:] + bounds[:-1]) / 2
    drift_x_pol = interpolate.InterpolatedUnivariateSpline(t, shift_x, k=3)
    drift_y_pol = interpolate.InterpolatedUnivariateSpline(t, shift_y, k=3)
    t_inter = np.arange(info[0]["Frames"])
    drift = pd.DataFrame({"x": drift_x_pol(t_inter), "y": drift_y_pol(t_inter)})
    locs = apply_drift(locs, info, drift=drift)
    return drift, locs
