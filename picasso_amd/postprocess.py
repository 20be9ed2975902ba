"""The RCC drift-correction entry points of picasso.postprocess (picasso/postprocess.py:2824-2961
``n_segments``, ``segment``, ``undrift``; :3157-3218 ``_apply_drift`` / ``apply_drift``) on top of
the GPU render (csrc/render.hip) and cross-correlation (csrc/xcorr.hip).
"""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from scipy.interpolate import InterpolatedUnivariateSpline

from . import backend, imageprocess, lib, render

_SEGMENT_RENDER = {"blur_method": "gaussian", "min_blur_width": 1}      # what undrift renders its segments with


def n_segments(info, segmentation: int) -> int:
    """Number of temporal segments: round(Frames / segmentation)."""
    return int(np.round(lib.get_from_metadata(info, "Frames") / segmentation))


def _segment_bounds(info, segmentation: int) -> np.ndarray:
    count = n_segments(info, segmentation)
    return np.linspace(0, info[0]["Frames"] - 1, count + 1, dtype=np.uint32)


def segment(locs: pd.DataFrame, info, segmentation: int, kwargs: dict = {}, callback=None):
    """Render the localizations of every temporal segment (postprocess.py:2846-2897).
    -> (uint32 frame bounds, float64 stack of shape (n_segments, Height, Width)); segment i holds the
    frames bounds[i] <= frame < bounds[i + 1]; the callback sees 0, 1, ..., n_segments."""
    bounds = _segment_bounds(info, segmentation)
    stack = np.zeros((len(bounds) - 1, info[0]["Height"], info[0]["Width"]))
    frame = locs["frame"]
    if callback is not None:
        callback(0)
    # A localization table is ordered by frame (picasso/gaussmle.py:1036): a segment is then a row range, found by
    # bisection, and only the columns a render reads are taken — instead of a boolean mask over every row and a copy
    # of all 17 columns per segment (4e7 rows x 25 segments on one rank of config 4: seconds).
    cols = [c for c in ("x", "y", "lpx", "lpy") if c in locs.columns]
    fr = frame.to_numpy()
    ordered = len(fr) == 0 or bool(np.all(fr[1:] >= fr[:-1]))
    views = {c: locs[c].to_numpy() for c in cols} if ordered else None
    for i, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        if ordered:
            a, b = np.searchsorted(fr, lo, side="left"), np.searchsorted(fr, hi, side="left")
            part = pd.DataFrame({c: views[c][a:b] for c in cols}, copy=False)
        else:
            part = locs[(frame >= lo) & (frame < hi)]
        stack[i] = render.render(part, info, **kwargs)[1]
        if callback is not None:
            callback(i + 1)
    return bounds, stack


def _apply_drift(locs: pd.DataFrame, drift: pd.DataFrame) -> pd.DataFrame:
    """coordinate -= drift[frame]; the float64 drift turns the float32 columns into float64, as in
    the reference (postprocess.py:3157-3168)."""
    row = locs["frame"].to_numpy()
    for axis in ("x", "y", "z"):
        if axis in drift.columns and axis in locs.columns:
            locs[axis] = locs[axis] - drift[axis].to_numpy()[row]
    return locs


def apply_drift(locs: pd.DataFrame, info, *, drift):
    """Checked form (postprocess.py:3171-3218): drift is a DataFrame with x, y (, z) per frame or an
    array of shape (Frames, 2 | 3)."""
    assert isinstance(drift, (pd.DataFrame, np.ndarray)), "Drift must be a DataFrame or numpy array"
    n_frames = lib.get_from_metadata(info, "Frames", raise_error=True)
    if isinstance(drift, np.ndarray):
        if drift.shape[0] != n_frames or drift.shape[1] not in (2, 3):
            raise ValueError("Drift array must have shape (n_frames, 2) for x and y drift, "
                             "or (n_frames, 3) for x, y, and z drift.")
        drift = pd.DataFrame(drift, columns=["x", "y", "z"][:drift.shape[1]])
    elif not {"x", "y"} <= set(drift.columns):
        raise ValueError(f"Drift DataFrame must contain columns {{'x', 'y'}}")
    return _apply_drift(locs, drift)


def _spline_over_frames(bounds, shift, n_frames):
    centres = (bounds[1:] + bounds[:-1]) / 2
    return InterpolatedUnivariateSpline(centres, shift, k=3)(np.arange(n_frames))


def undrift(locs: pd.DataFrame, info, segmentation: int, display: bool = True, segmentation_callback=None,
            rcc_callback=None):
    """RCC drift correction (postprocess.py:2900-2961) -> (drift DataFrame, undrifted copy of locs).
    ``display`` (a matplotlib plot in the reference) is ignored."""
    locs = locs.copy(deep=False)          # the drift replaces the x / y columns of the copy; the others are shared
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)       # render()'s oversampling notice, as in the reference call
        bounds, stack = segment(locs, info, segmentation, dict(_SEGMENT_RENDER), segmentation_callback)
    backend.join_fft_prewarm()            # plans started by localize_file on a side thread: ready (or made) before they are used
    shift_y, shift_x = imageprocess.rcc(stack, 32, rcc_callback)
    n_frames = info[0]["Frames"]
    drift = pd.DataFrame({"x": _spline_over_frames(bounds, shift_x, n_frames),
                          "y": _spline_over_frames(bounds, shift_y, n_frames)})
    return drift, apply_drift(locs, info, drift=drift)
