"""picasso.localize surface on the HIP backend.

Drop-in for the hot-path functions of picasso/localize.py: ``identify``
(:639-749), ``identify_by_frame_number`` (:340-421), ``identify_in_frame``
(:295-337), ``identify_in_image`` (:247-292), ``get_spots`` (:1115-1145),
``fit2D`` (:1344-1506), ``localize`` (:1682-1815) and ``localize_3D`` (:1818-2031) — same signatures,
defaults, return types, metadata dictionaries, assertion messages and
abort/progress contracts — plus the older names the module still exports (``fit``, ``fit_async``,
``identify_async``, ``identifications_from_futures``, ``locs_from_fits``, ``local_maxima``, ``net_gradient``, ``gradient_at``) and the
identification constructors the GUI feeds into ``get_spots`` (``picks_to_identifications``,
``locs_to_identifications``).  The work is done by libpicasso_hip.so; there is no
CPU path.  ``install()`` rebinds the reference package's functions to these.
"""
from __future__ import annotations

import time
import warnings
from typing import Callable, Literal

import numpy as np
import pandas as pd

from . import __version__, _lib, avgroi, backend, gausslq, gaussmle, lib

_CHUNK_BYTES = 1 << 30      # movie bytes per device call = progress / abort granularity
FITTING_METHODS = ["gausslq", "gausslq-gpu", "gaussmle", "avg"]


def _deprecation_warning(message: str) -> None:
    warnings.warn(message, DeprecationWarning, stacklevel=3)


# ---------------------------------------------------------------------------
# movie access
# ---------------------------------------------------------------------------
def _is_array_movie(movie) -> bool:
    return isinstance(movie, np.ndarray)          # includes np.memmap (.raw movies)


def _accepted_movie(movie) -> bool:
    """What picasso.io.load_movie returns: a memmap or an AbstractPicassoMovie
    (duck-typed here: len(), [i] -> 2-D frame, .dtype)."""
    if isinstance(movie, np.memmap):
        return True
    if isinstance(movie, np.ndarray):
        return False                               # reference rejects bare ndarrays (localize.py:1416-1424)
    return hasattr(movie, "__len__") and hasattr(movie, "__getitem__") and hasattr(movie, "dtype")


def _frames_block(movie, f0: int, f1: int) -> np.ndarray:
    """Frames [f0, f1) as one C-contiguous array (zero-copy for ndarrays)."""
    if _is_array_movie(movie):
        return backend.as_movie_array(movie[f0:f1])
    return backend.as_movie_array(np.stack([np.asarray(movie[i]) for i in range(f0, f1)]))


def _movie_shape(movie):
    n = len(movie)
    if _is_array_movie(movie):
        return n, movie.shape[1], movie.shape[2]
    f = np.asarray(movie[0]) if n else np.zeros((0, 0))
    return n, f.shape[0], f.shape[1]


def _chunk_frames(movie) -> int:
    n, Y, X = _movie_shape(movie)
    per_frame = max(1, Y * X * np.dtype(movie.dtype).itemsize)
    return max(1, _CHUNK_BYTES // per_frame)


def _empty_identifications() -> pd.DataFrame:
    return pd.DataFrame({"frame": pd.Series(dtype=int), "x": pd.Series(dtype=int), "y": pd.Series(dtype=int),
                         "net_gradient": pd.Series(dtype=np.float32)})


def _ids_frame(fr, y, x, ng) -> pd.DataFrame:
    # column order and dtypes of localize.py:413-420
    return pd.DataFrame({"frame": fr.astype(int), "x": x.astype(int), "y": y.astype(int),
                         "net_gradient": ng.astype(np.float32)})


# ---------------------------------------------------------------------------
# identify
# ---------------------------------------------------------------------------
def identify_in_image(image, minimum_ng: float, box: int):
    """Local maxima + net gradient of one float32 image -> y, x, ng."""
    img = np.ascontiguousarray(image, dtype=np.float32)[None]
    fr, y, x, ng = backend.identify_arrays(img, minimum_ng, box)
    return y.astype(np.int64), x.astype(np.int64), ng


def identify_in_frame(frame, minimum_ng: float, box: int, roi=None):
    """One frame, optional ROI ((y0, x0), (y1, x1)); coordinates are frame coordinates."""
    fr, y, x, ng = backend.identify_arrays(np.asarray(frame)[None], minimum_ng, box, roi=roi)
    return y.astype(np.int64), x.astype(np.int64), ng


def identify_by_frame_number(movie, minimum_ng: float, box: int, frame_number: int, *, roi=None,
                             frame_bounds=None, lock=None) -> pd.DataFrame:
    if lock is not None:
        with lock:
            frame = movie[frame_number]
    else:
        frame = movie[frame_number]
    lo, hi = backend.frame_range(frame_bounds, len(movie))
    if frame_bounds is not None and not (lo <= frame_number <= hi):
        return _empty_identifications()
    y, x, ng = identify_in_frame(frame, minimum_ng, box, roi)
    return _ids_frame(frame_number * np.ones(len(x)), y, x, ng)


def identify(movie, minimum_ng: float, box: int, *, roi=None, frame_bounds=None, threaded: bool = True,
             progress_callback=None, abort_callback=None, return_info: bool = None):
    """Identify spots in every frame (picasso/localize.py:639-749).

    ``threaded`` is accepted for compatibility (the device call is one batch per
    ~1 GiB of frames either way); ``progress_callback`` receives the number of
    frames done, ``"console"`` shows a tqdm bar; when ``abort_callback()`` turns
    true between batches the function returns ``None`` like the reference.
    """
    if return_info is None:
        return_info = False
        _deprecation_warning(
            "Warning: In Picasso v0.11.0, picasso.localize.identify() will return both the identifications "
            "and a metadata dictionary by default.\nBefore v0.12.0, when using picasso.localize.identify(), "
            "please add the argument 'return_info' explicitly as True or False.\nIn version 0.12, this "
            "argument will also be removed such that picasso.localize.identify() will always return both "
            "the identifications and the metadata dictionary.")
    N = len(movie)
    lo, hi = backend.frame_range(frame_bounds, N)
    hi = min(hi, N - 1)
    bar = None
    if progress_callback == "console":
        from tqdm import tqdm
        bar = tqdm(total=N, desc="Identifying spots", unit="frame")
    parts = []
    step = _chunk_frames(movie) if N else 1
    f0 = max(lo, 0)
    while f0 <= hi:
        if threaded and abort_callback is not None and abort_callback():
            if bar is not None:
                bar.close()
            return None
        f1 = min(hi + 1, f0 + step)
        block = _frames_block(movie, f0, f1)
        fr, y, x, ng = backend.identify_arrays(block, minimum_ng, box, roi=roi)
        parts.append(_ids_frame(fr.astype(np.int64) + f0, y, x, ng))
        if bar is not None:
            bar.update(f1 - f0)
        elif callable(progress_callback):
            progress_callback(f1 if threaded else f1 - 1)
        f0 = f1
    if bar is not None:
        bar.update(N - bar.n)
        bar.close()
    ids = pd.concat(parts, ignore_index=True) if parts else _empty_identifications()
    if return_info:
        info = {"Generated by": f"Picasso: v{__version__} Identify",
                "Min. Net Gradient": minimum_ng, "Box Size": box, "ROI": roi, "Frame Bounds": frame_bounds}
        return ids, info
    return ids


# ---------------------------------------------------------------------------
# spots
# ---------------------------------------------------------------------------
def get_spots(movie, identifications: pd.DataFrame, box: int, camera_info: dict) -> np.ndarray:
    """ROI extraction + camera-signal -> photon conversion (localize.py:1115-1145)."""
    frame = identifications["frame"].to_numpy().astype(np.int64)
    y = identifications["y"].to_numpy().astype(np.int64)      # picks-derived ids may be floats: truncation like int indexing
    x = identifications["x"].to_numpy().astype(np.int64)
    N = len(frame)
    cam = (camera_info["Baseline"], camera_info["Sensitivity"], camera_info["Gain"])
    if N == 0:
        return np.zeros((0, box, box), np.float32)
    if _is_array_movie(movie):
        return backend.get_spots_array(movie, frame, y, x, box, *cam)
    # frame-by-frame movies: upload runs of frames (assumes frame-ordered ids, as the reference does)
    spots = np.empty((N, box, box), np.float32)
    step = _chunk_frames(movie)
    order = np.argsort(frame, kind="stable")
    fs = frame[order]
    i0 = 0
    while i0 < N:
        f0 = int(fs[i0])
        i1 = int(np.searchsorted(fs, f0 + step, side="left"))
        f1 = int(fs[i1 - 1]) + 1
        block = _frames_block(movie, f0, f1)
        sel = order[i0:i1]
        spots[sel] = backend.get_spots_array(block, frame[sel] - f0, y[sel], x[sel], box, *cam)
        i0 = i1
    return spots


# ---------------------------------------------------------------------------
# fit
# ---------------------------------------------------------------------------
def _fit2d_gaussmle(spots, identifications, box, eps=0.001, max_it=100, mle_method="sigmaxy", multiprocess=True,
                    progress_callback=None, abort_callback=None):
    N = len(identifications)
    bar = None
    if progress_callback == "console":
        from tqdm import tqdm
        bar = tqdm(total=N, desc="Fitting", unit="spot")
    if multiprocess:
        curr, thetas, CRLBs, llhoods, iterations = gaussmle.gaussmle_async(spots, eps, max_it, method=mle_method)
        last = 0
        while curr[0] < N:
            if callable(abort_callback) and abort_callback():
                if bar is not None:
                    bar.close()
                return None
            if bar is not None:
                bar.update(curr[0] - last)
                last = curr[0]
            elif callable(progress_callback):
                progress_callback(curr[0])
            time.sleep(0.02)
        gaussmle.wait_async(curr)
        if bar is not None:
            bar.update(N - last)
            bar.close()
    else:
        thetas, CRLBs, llhoods, iterations = gaussmle.gaussmle(spots, eps, max_it, mle_method, progress_callback)
    return gaussmle.locs_from_fits(identifications, thetas, CRLBs, llhoods, iterations, box)


def _fit2d_gausslq(spots, identifications, box, em, multiprocess=True, progress_callback=None, abort_callback=None):
    """picasso/localize.py:1509-1538.  The fit is one kernel launch, so the abort
    callback is polled once before it."""
    if callable(abort_callback) and abort_callback():
        return None
    theta = gausslq.fit_spots(spots, progress_callback if callable(progress_callback) else None)
    return gausslq.locs_from_fits(identifications, theta, box, em)


def fit2D(movie, movie_info, camera_info: dict, identifications: pd.DataFrame, box: int,
          fitting_method: Literal["gausslq", "gausslq-gpu", "gaussmle", "avg"] = "gausslq", eps: float = 0.001,
          max_it: int = 100, mle_method: Literal["sigma", "sigmaxy"] = "sigmaxy", multiprocess: bool = True,
          progress_callback=None, abort_callback=None):
    """Fit 2D localizations (picasso/localize.py:1344-1506).  Returns (locs | None, new_info)."""
    assert _accepted_movie(movie), "movie must be a movie loaded by picasso.io.load_movie"
    assert isinstance(movie_info, list), "movie_info must be a list"
    assert isinstance(camera_info, dict), "camera_info must be a dict"
    assert isinstance(identifications, pd.DataFrame), "identifications must be a DataFrame"
    assert isinstance(box, int) and box > 0, "box must be a positive integer"
    assert fitting_method in FITTING_METHODS, (
        "fitting_method must be one of 'gausslq', 'gausslq-gpu', 'gaussmle', or 'avg'")
    assert isinstance(eps, (int, float)) and eps > 0, "eps must be a positive number"
    assert isinstance(max_it, int) and max_it > 0, "max_it must be a positive integer"
    assert mle_method in ["sigma", "sigmaxy"], "mle_method must be 'sigma' or 'sigmaxy'"
    assert isinstance(multiprocess, bool), "multiprocess must be a boolean"
    if "Pixelsize" not in camera_info:
        warnings.warn("Camera info in picasso.localize.fit2D does not contain 'Pixelsize', i.e., effective "
                      "camera pixel size in nm. Assuming 130.")
        camera_info["Pixelsize"] = 130

    if fitting_method == "gausslq-gpu":
        raise NotImplementedError(
            "fitting_method='gausslq-gpu' is the reference's CUDA Gpufit binding (a different, float32 "
            "Levenberg-Marquardt); picasso_amd has no equivalent — 'gausslq' already runs on the GPU here")
    spots = get_spots(movie, identifications, box, camera_info)
    em = camera_info["Gain"] > 1
    if fitting_method == "gausslq":
        locs = _fit2d_gausslq(spots, identifications, box, em, multiprocess, progress_callback, abort_callback)
    elif fitting_method == "gaussmle":
        locs = _fit2d_gaussmle(spots, identifications, box, eps, max_it, mle_method, multiprocess,
                               progress_callback, abort_callback)
    else:
        if callable(abort_callback) and abort_callback():
            locs = None
        else:
            locs = avgroi.locs_from_fits(identifications, avgroi.fit_spots(spots, progress_callback), box, em)
    localize_info = {"Generated by": f"Picasso: v{__version__} Fit 2D",
                     "Fit method": fitting_method}
    if fitting_method == "gaussmle":
        localize_info["Convergence criterion"] = eps
        localize_info["Max iterations"] = max_it
    return locs, localize_info | camera_info


def localize(movie, camera_info: dict, parameters: dict, *, roi=None, frame_bounds=None, movie_info=None,
             fitting_method: Literal["gausslq", "gausslq-gpu", "gaussmle", "avg"] = "gausslq", eps: float = 0.001,
             max_it: int = 100, mle_method: Literal["sigma", "sigmaxy"] = "sigmaxy", threaded: bool = True,
             identification_progress_callback=None, fit_progress_callback=None, return_info: bool = None):
    """identify + fit2D (picasso/localize.py:1682-1815)."""
    if return_info is None:
        return_info = False
        _deprecation_warning(
            "Warning: In Picasso v0.11.0, picasso.localize.localize() will return both the localizations and "
            "a metadata dictionary by default.\nBefore v0.12.0, when using picasso.localize.localize(), please "
            "add the argument 'return_info' explicitly as True or False.\nIn version 0.12, this argument will "
            "also be removed such that picasso.localize.localize() will always return both the localizations "
            "and the metadata dictionary.")
    if movie_info is None:
        movie_info = []
    if fitting_method in ("gaussmle", "gausslq"):
        # one upload per frame chunk, identify -> cut -> fit -> table on the device: the same table as the
        # two calls below, without sending the movie over PCIe twice or the identifications back and forth
        box, min_ng = parameters["Box Size"], parameters["Min. Net Gradient"]
        locs = localize_streamed(movie, camera_info, parameters, roi=roi, frame_bounds=frame_bounds,
                                 fitting_method=fitting_method, eps=eps, max_it=max_it, mle_method=mle_method,
                                 progress_callback=identification_progress_callback)
        if callable(fit_progress_callback):
            fit_progress_callback(len(locs), len(locs))
        identify_info = {"Generated by": f"Picasso: v{__version__} Identify",
                         "Min. Net Gradient": min_ng, "Box Size": box, "ROI": roi, "Frame Bounds": frame_bounds}
        fit_info = {"Generated by": f"Picasso: v{__version__} Fit 2D",
                    "Fit method": fitting_method}
        if fitting_method == "gaussmle":
            fit_info["Convergence criterion"] = eps
            fit_info["Max iterations"] = max_it
        info = movie_info + [identify_info] + [fit_info | camera_info]
        return (locs, info) if return_info else locs
    identifications, identify_info = identify(movie, parameters["Min. Net Gradient"], parameters["Box Size"],
                                              roi=roi, frame_bounds=frame_bounds, threaded=threaded,
                                              progress_callback=identification_progress_callback,
                                              return_info=True)
    locs, fit_info = fit2D(movie=movie, movie_info=movie_info, camera_info=camera_info,
                           identifications=identifications, box=parameters["Box Size"],
                           fitting_method=fitting_method, eps=eps, max_it=max_it, mle_method=mle_method,
                           multiprocess=threaded, progress_callback=fit_progress_callback)
    info = movie_info + [identify_info] + [fit_info]
    if return_info:
        return locs, info
    return locs


# ---------------------------------------------------------------------------
# the remaining public names of picasso/localize.py on the path
# ---------------------------------------------------------------------------
def _local_maxima(frame, box: int):
    """y, x of the pixels that are the first maximum of their box x box window (picasso/localize.py:97-134),
    in np.where order."""
    img = np.ascontiguousarray(frame, dtype=np.float32)[None]
    _, y, x, _ = backend.identify_arrays(img, -np.inf, box)
    return y.astype(np.int64), x.astype(np.int64)


def local_maxima(frame, box: int):
    """Older public name of ``_local_maxima`` (picasso/localize.py:84-94)."""
    _deprecation_warning("Deprecation warning: This function will become private in v0.11.0. "
                         "Use _local_maxima instead.")
    return _local_maxima(frame, box)


def _net_gradient(frame, y, x, box: int, uy, ux):
    """Net gradient (float32) at pixels (y, x) of one frame with the given unit-vector tables
    (picasso/localize.py:202-244).  The frame is taken as float32, as ``identify_in_frame`` hands it over."""
    return backend.net_gradient_array(frame, y, x, box, uy, ux)


def net_gradient(frame, y, x, box: int, uy, ux):
    """Older public name of ``_net_gradient`` (picasso/localize.py:184-199)."""
    _deprecation_warning("Deprecation warning: This function will become private in v0.11.0. "
                         "Use _net_gradient instead.")
    return _net_gradient(frame, y, x, box, uy, ux)


def _gradient_at(frame, y: int, x: int, i: int = 0):
    """Central differences (gy, gx) at one pixel (picasso/localize.py:153-181) — two subtractions in the frame's
    own dtype, as indexing arithmetic on the host."""
    return frame[y + 1, x] - frame[y - 1, x], frame[y, x + 1] - frame[y, x - 1]


def gradient_at(frame, y: int, x: int, i: int = 0):
    """Older public name of ``_gradient_at`` (picasso/localize.py:137-150)."""
    _deprecation_warning("Deprecation warning: This function will become private in v0.11.0. "
                         "Use _gradient_at instead.")
    return _gradient_at(frame, y, x, i)


def identify_async(movie, minimum_ng: float, box: int, *, roi=None, frame_bounds=None):
    """picasso/localize.py:482-560 returns a frame counter and one future per worker thread; here the device
    call has finished when this returns: the counter stands at the last frame and the single future holds the
    per-chunk identification frames ``identifications_from_futures`` expects."""
    ids = identify(movie, minimum_ng, box, roi=roi, frame_bounds=frame_bounds, threaded=True, return_info=False)
    return [len(movie)], [gausslq._DoneFuture([ids])]


def identifications_from_futures(futures) -> pd.DataFrame:
    """Concatenate the workers' lists of per-frame identifications, ordered by frame (picasso/localize.py:457-479)."""
    frames = [df for f in futures for df in f.result()]
    ids = pd.concat(frames, ignore_index=True) if frames else _empty_identifications()
    return ids.sort_values(by="frame", kind="stable")


def locs_from_fits(identifications: pd.DataFrame, theta, CRLBs, likelihoods, iterations, box: int) -> pd.DataFrame:
    """The older 12-column table of picasso/localize.py:1281-1341 (columns ``likelihood`` and int32
    ``iterations``; theta is read as (y, x, photons, bg, sy, sx) and no box offset is subtracted).  The
    current builder is ``gaussmle.locs_from_fits``."""
    idf = identifications
    y = theta[:, 0] + idf["y"].to_numpy()
    x = theta[:, 1] + idf["x"].to_numpy()
    locs = pd.DataFrame({
        "frame": idf["frame"].to_numpy().astype(np.uint32), "x": x.astype(np.float32), "y": y.astype(np.float32),
        "photons": theta[:, 2].astype(np.float32), "sx": theta[:, 5].astype(np.float32),
        "sy": theta[:, 4].astype(np.float32), "bg": theta[:, 3].astype(np.float32),
        "lpx": np.sqrt(CRLBs[:, 1]).astype(np.float32), "lpy": np.sqrt(CRLBs[:, 0]).astype(np.float32),
        "net_gradient": idf["net_gradient"].to_numpy().astype(np.float32),
        "likelihood": np.asarray(likelihoods).astype(np.float32), "iterations": np.asarray(iterations).astype(np.int32)})
    return locs.sort_values(by="frame", kind="stable")


def fit(movie, camera_info: dict, identifications: pd.DataFrame, box: int, eps: float = 0.001, max_it: int = 100,
        method: Literal["sigma", "sigmaxy"] = "sigmaxy") -> pd.DataFrame:
    """Older MLE entry point (picasso/localize.py:1148-1211): spots -> gaussmle -> ``locs_from_fits``.  (At the
    reference's HEAD the call into ``locs_from_fits`` drops the CRLB argument and raises; this passes all six.)"""
    _deprecation_warning("Deprecation warning: this function will be removed in v0.11.0. Use localize.fit2D instead.")
    spots = get_spots(movie, identifications, box, camera_info)
    theta, CRLBs, likelihoods, iterations = gaussmle.gaussmle(spots, eps, max_it, method=method)
    return locs_from_fits(identifications, theta, CRLBs, likelihoods, iterations, box)


def fit_async(movie, camera_info: dict, identifications: pd.DataFrame, box: int, eps: float = 0.001,
              max_it: int = 100, method: Literal["sigma", "sigmaxy"] = "sigmaxy"):
    """picasso/localize.py:1214-1278: the counter and result arrays of ``gaussmle.gaussmle_async``."""
    _deprecation_warning("Deprecation warning: this function will be removed in v0.11.0. Use localize.fit2D instead.")
    spots = get_spots(movie, identifications, box, camera_info)
    return gaussmle.gaussmle_async(spots, eps, max_it, method=method)


def _ids_around(frames, xs, ys, n_ids) -> pd.DataFrame:
    """frame / x / y / net_gradient (the dummy 101) / n_id, all float64 as the reference builds them from a
    float array, ordered by frame."""
    ids = pd.DataFrame({"frame": np.asarray(frames, float), "x": np.asarray(xs, float), "y": np.asarray(ys, float),
                        "net_gradient": np.full(len(frames), 101.0), "n_id": np.asarray(n_ids, float)})
    return ids.sort_values(by="frame", kind="stable")


def picks_to_identifications(picks, *, n_frames: int | None = None, drift: pd.DataFrame | None = None) -> pd.DataFrame:
    """Circular picks -> one identification per pick and frame, following the drift when given
    (picasso/localize.py:752-854); n_id counts picks from 1."""
    assert isinstance(picks, (list, tuple)), "picks must be a list or a tuple."
    assert all([len(_) == 2 for _ in picks]), (
        "Circular picks are required. Each element in 'picks' must contain two numbers (x and y coordinates).")
    if isinstance(drift, pd.DataFrame):
        assert all(col in drift.columns for col in ["x", "y"]), "Drift data frame must contain 'x' and 'y' columns."
    if n_frames is None:
        if drift is None:
            raise ValueError("n_frames must be given if no drift file is provided")
        n_frames = len(drift)
    else:
        assert isinstance(n_frames, int), "n_frames must be an integer."
        if drift is not None:
            assert n_frames == len(drift), (
                f"{n_frames} frames were provided but the drift suggests {len(drift)} frames.")
    n_picks = len(picks)
    centres = np.asarray(picks, float).reshape(n_picks, 2)
    dx = drift["x"].to_numpy() if drift is not None else np.zeros(n_frames)
    dy = drift["y"].to_numpy() if drift is not None else np.zeros(n_frames)
    frames = np.tile(np.arange(n_frames), n_picks)
    xs = (centres[:, :1] + dx[None, :]).ravel()
    ys = (centres[:, 1:] + dy[None, :]).ravel()
    return _ids_around(frames, xs, ys, np.repeat(np.arange(n_picks) + 1.0, n_frames))


def locs_to_identifications(locs: pd.DataFrame, movie_info, n_frames: int) -> pd.DataFrame:
    """Each localization -> identifications at its position over the 2 n_frames + 1 frames around it
    (picasso/localize.py:857-913); localizations closer than n_frames to either end are skipped but still
    counted in n_id."""
    assert isinstance(locs, pd.DataFrame), "Localizations must be a pandas data frame"
    assert isinstance(n_frames, int) and n_frames >= 0, "n_frames must be a non-negative integer"
    max_frames = lib.get_from_metadata(movie_info, "Frames", raise_error=True)
    f = locs["frame"].to_numpy().astype(float)
    keep = np.nonzero((f > n_frames) & (f < max_frames - n_frames))[0]
    span = np.arange(-n_frames, n_frames + 1, dtype=float)
    frames = (f[keep, None] + span[None, :]).ravel()
    k = len(span)
    return _ids_around(frames, np.repeat(locs["x"].to_numpy().astype(float)[keep], k),
                       np.repeat(locs["y"].to_numpy().astype(float)[keep], k), np.repeat(keep + 1.0, k))


def localize_3D(movie, *, movie_info, camera_info: dict, box: int, minimum_ng: float, calibration_3d,
                roi=None, frame_bounds=None, fitting_method: Literal["gausslq", "gausslq-gpu", "gaussmle"] = "gausslq",
                eps: float = 0.001, max_it: int = 100, mle_method: Literal["sigma", "sigmaxy"] = "sigmaxy",
                multiprocess: bool = True, identification_progress_callback=None, fit_progress_callback=None,
                fit_z_progress_callback=None):
    """2D localization followed by the astigmatic z fit (picasso/localize.py:1818-1975) -> (locs, info)."""
    assert isinstance(movie, np.ndarray) or _accepted_movie(movie), "movie must be a numpy array or ND2Movie"
    assert isinstance(movie_info, list), "movie_info must be a list"
    assert isinstance(camera_info, dict), "camera_info must be a dict"
    assert isinstance(box, int) and box > 0 and box % 2 == 1, "box must be a positive odd integer"
    assert isinstance(minimum_ng, (int, float)), "minimum_ng must be a number"
    assert isinstance(calibration_3d, (dict, str)), "calibration_3d must be a dict or a path to a YAML file"
    assert fitting_method in ["gausslq", "gausslq-gpu", "gaussmle"], (
        "fitting_method must be one of 'gausslq', 'gausslq-gpu', or 'gaussmle'")
    assert isinstance(eps, (int, float)) and eps > 0, "eps must be a positive number"
    assert isinstance(max_it, int) and max_it > 0, "max_it must be a positive integer"
    assert mle_method in ["sigma", "sigmaxy"], "mle_method must be 'sigma' or 'sigmaxy'"
    assert isinstance(multiprocess, bool), "multiprocess must be a boolean"
    return _localize_3D(movie, movie_info=movie_info, camera_info=camera_info, box=box, minimum_ng=minimum_ng,
                        calibration_3d=calibration_3d, roi=roi, frame_bounds=frame_bounds,
                        fitting_method=fitting_method, eps=eps, max_it=max_it, mle_method=mle_method,
                        multiprocess=multiprocess, identification_progress_callback=identification_progress_callback,
                        fit_progress_callback=fit_progress_callback, fit_z_progress_callback=fit_z_progress_callback)


def _localize_3D(movie, *, movie_info, camera_info, box, minimum_ng, calibration_3d, roi=None, frame_bounds=None,
                 fitting_method="gausslq", eps=0.001, max_it=100, mle_method="sigmaxy", multiprocess=True,
                 identification_progress_callback=None, fit_progress_callback=None, fit_z_progress_callback=None):
    """picasso/localize.py:1977-2031."""
    from . import zfit
    locs, info = localize(movie, camera_info, {"Min. Net Gradient": minimum_ng, "Box Size": box}, roi=roi,
                          frame_bounds=frame_bounds, movie_info=movie_info, fitting_method=fitting_method, eps=eps,
                          max_it=max_it, mle_method=mle_method, threaded=multiprocess,
                          identification_progress_callback=identification_progress_callback,
                          fit_progress_callback=fit_progress_callback, return_info=True)
    method_3d = "gausslq" if fitting_method in ["gausslq", "gausslq-gpu"] else "gaussmle"
    return zfit.zfit(locs=locs, info=info, calibration=calibration_3d, fitting_method=method_3d, filter=0,
                     multiprocess=multiprocess, progress_callback=fit_z_progress_callback)


def localize_resident(movie: np.ndarray, camera_info: dict, parameters: dict, *, roi=None, frame_bounds=None,
                      fitting_method: str = "gaussmle", eps: float = 0.001, max_it: int = 100,
                      mle_method: str = "sigmaxy") -> pd.DataFrame:
    """The fused device pipeline (identify -> cut+fit -> table, one submission, no
    host round trip) for a movie that fits in HBM.  Same table as ``localize`` with
    the same ``fitting_method`` ("gaussmle" or "gausslq"); the gaussmle form is what
    bench.py times."""
    if fitting_method not in ("gaussmle", "gausslq"):
        raise ValueError("localize_resident supports fitting_method 'gaussmle' or 'gausslq'")
    dm = backend.DeviceMovie(movie)
    try:
        if fitting_method == "gausslq":
            cols = backend.localize_lq_device(dm.ptr, dm.dtype, dm.shape, parameters["Box Size"],
                                              parameters["Min. Net Gradient"], camera_info, roi=roi,
                                              frame_bounds=frame_bounds)
        else:
            cols = backend.localize_mle_device(dm.ptr, dm.dtype, dm.shape, parameters["Box Size"],
                                               parameters["Min. Net Gradient"], camera_info, eps, max_it, mle_method,
                                               roi=roi, frame_bounds=frame_bounds)
    finally:
        dm.free()
    return pd.DataFrame(cols)


class _DeviceLane:
    """One lane of a streamed run: a device (and one of its two scratch banks), two staging allocations for the frame
    chunks, a non-blocking stream and the table buffers.  `bind` runs first in every host thread that works for the lane
    (its uploader and its worker): the library keys what it keeps on a device by the calling thread's device."""

    def __init__(self, device, bank, call):
        self.device, self.bank, self._call = device, bank, call
        self.stages = [None, None]
        self.stream = None
        self.work = None

    def bind(self):
        _lib.bind_thread(self.device, self.bank)

    def open(self):
        self.stream = backend.DeviceStream()
        self.work = backend.DeviceWorkspace()

    def upload(self, k: int, chunk: np.ndarray):
        if self.stages[k] is None:
            self.stages[k] = backend.DeviceMovie(chunk)
        else:
            self.stages[k].load(chunk)

    def run(self, k: int, c0: int) -> pd.DataFrame:
        cols = self._call(self.stages[k], self.stream.handle, self.work)
        cols["frame"] = cols["frame"] + np.asarray(c0, cols["frame"].dtype)
        return pd.DataFrame(cols)

    def close(self):
        for st in self.stages:
            if st is not None:
                st.free()
        if self.work is not None:
            self.work.free()
        if self.stream is not None:
            self.stream.destroy()


def _run_lanes(movie, chunks, lanes, progress_callback=None, abort_callback=None, frames=None):
    """The scheduler of `localize_streamed`: chunk i = frames [c0, c1) goes to lane i % len(lanes).  Every lane has one
    host thread that uploads the lane's next chunk — a blocking copy, the GIL
    released — while a worker thread of the lane runs identify -> cut + fit -> table on the previous one in the lane's other
    staging allocation.  Tables come back in frame order whatever order the lanes finish in (what the reference's worker
    threads give, picasso/localize.py:424-454 + the sort at :478).  `progress_callback(n)`: frames handed to a device so
    far, from whichever lane thread got there; `abort_callback()` is asked before every chunk: the run then stops, the
    chunks in flight finish, and None is returned (identify's contract, picasso/localize.py:462-470).
    A lane needs: bind(), open(), upload(k, chunk), run(k, c0) -> DataFrame, close()."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    frames = frames or _frames
    results = [None] * len(chunks)
    done_frames = [0]
    mu = threading.Lock()
    stop = threading.Event()
    errors = []

    def lane_loop(li, lane):
        mine = [(i, c) for i, c in enumerate(chunks) if i % len(lanes) == li]
        opened = False
        try:
            lane.bind()
            lane.open()
            opened = True
            futures = []
            with ThreadPoolExecutor(max_workers=1, initializer=lane.bind) as pool:
                for n, (i, (c0, c1)) in enumerate(mine):
                    if stop.is_set():
                        break
                    if callable(abort_callback) and abort_callback():
                        stop.set()
                        break
                    if n >= 2:
                        futures[n - 2][1].result()      # that chunk is done with the staging allocation this one takes
                    lane.upload(n & 1, frames(movie, c0, c1))
                    futures.append((i, pool.submit(lane.run, n & 1, c0)))
                    if callable(progress_callback):
                        with mu:
                            done_frames[0] += c1 - c0
                            progress_callback(done_frames[0])
                for i, f in futures:
                    results[i] = f.result()
        except BaseException as exc:      # noqa: BLE001 - handed to the calling thread
            errors.append(exc)
            stop.set()
        finally:
            if opened:
                lane.close()

    # every lane on a thread of its own, one lane too: binding a thread to a device and bank is sticky, and the calling
    # thread's device, bank and lock key must be what they were when this returns
    threads = [threading.Thread(target=lane_loop, args=(li, lane), name=f"pmi-lane-{li}") for li, lane in enumerate(lanes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    if stop.is_set():
        return None
    return [r for r in results if r is not None]


def localize_streamed(movie, camera_info: dict, parameters: dict, *, roi=None, frame_bounds=None,
                      fitting_method: str = "gaussmle", eps: float = 0.001, max_it: int = 100,
                      mle_method: str = "sigmaxy", chunk_bytes: int = 1 << 28,
                      progress_callback=None, abort_callback=None, devices=None) -> pd.DataFrame:
    """The fused device pipeline over a host movie of any length (ndarray, memmap or a picasso movie
    object): frames go up in chunks of about ``chunk_bytes`` through two staging allocations (the upload of
    one chunk overlaps the device work and the row copy of the previous one, which run on a stream of their
    own), each chunk runs identify -> cut+fit -> table on the device, only the table rows come back.  The movie crosses PCIe once
    and neither host RAM nor HBM has to hold it whole.  Same rows as ``localize_resident``.

    ``devices``: the GPUs of this process to use, e.g. ``[0, 1, 2, 3]`` or ``"all"`` — one host thread (+ its worker) per
    entry, the chunks dealt round robin, the tables concatenated in frame order; a device named twice gets two lanes on its
    two scratch banks.  This is how ONE process — the unmodified CLI / GUI after ``install()`` — reaches several GPUs and
    their PCIe links, as the reference reaches several cores with threads (picasso/localize.py:424-454).  ``None``: the
    calling thread's current device, as before.  ``abort_callback``: see ``_run_lanes``; returns None when it fired."""
    if fitting_method not in ("gaussmle", "gausslq"):
        raise ValueError("localize_streamed supports fitting_method 'gaussmle' or 'gausslq'")
    box, min_ng = parameters["Box Size"], parameters["Min. Net Gradient"]
    N = len(movie)
    lo, hi = backend.frame_range(frame_bounds, N)
    hi = min(hi, N - 1)
    columns = backend.LQ_COLUMNS if fitting_method == "gausslq" else backend.LOC_COLUMNS
    parts = []
    if hi >= lo:
        first = np.asarray(movie[lo])
        per = max(1, int(chunk_bytes) // max(first.nbytes, 1))
        chunks = [(c0, min(hi + 1, c0 + per)) for c0 in range(lo, hi + 1, per)]

        def call(stage, stream, work):
            if fitting_method == "gausslq":
                return backend.localize_lq_device(stage.ptr, stage.dtype, stage.shape, box, min_ng, camera_info, roi=roi,
                                                  stream=stream, work=work)
            return backend.localize_mle_device(stage.ptr, stage.dtype, stage.shape, box, min_ng, camera_info,
                                               eps, max_it, mle_method, roi=roi, stream=stream, work=work)

        lanes = [_DeviceLane(dev, bank, call) for dev, bank in _lanes_for(_resolve_devices(devices), len(chunks))]
        parts = _run_lanes(movie, chunks, lanes, progress_callback, abort_callback)
        if parts is None:
            return None
    if not parts:
        return pd.DataFrame({name: np.empty(0, dt) for name, dt in columns})
    return parts[0] if len(parts) == 1 else pd.concat(parts, ignore_index=True)


_default_devices = None       # set_devices(): what localize_streamed(devices=None) uses


def set_devices(devices) -> None:
    """The GPUs `localize_streamed` — and through it `localize()` and `localize_file()` — uses when no `devices` argument
    is given: None (the calling thread's current device), "all", or a list of device indices.  The environment variable
    PICASSO_AMD_DEVICES ("all" or "0,1,2,3") sets the same default for a process nobody can pass arguments to (the
    reference's CLI after `install()`)."""
    global _default_devices
    if devices is not None and not isinstance(devices, str):
        devices = [int(d) for d in devices]
    _default_devices = devices


def _resolve_devices(devices):
    if devices is not None:
        return devices
    if _default_devices is not None:
        return _default_devices
    import os
    env = os.environ.get("PICASSO_AMD_DEVICES", "").strip()
    if not env:
        return None
    return "all" if env == "all" else [int(t) for t in env.split(",") if t.strip()]


def _lanes_for(devices, n_chunks: int):
    """(device, scratch bank) of every lane.  None -> one lane on the calling thread's device and bank (what it bound
    itself to, else HIP's current device of the thread, bank 0) — named explicitly, so the lane's threads bind to it;
    "all" -> every visible device; a list -> as given, a device named twice on banks 0 and 1 (a third time is an error:
    the library has two banks per device)."""
    if devices is None:
        key = _lib.current_key()
        if key is None:
            _lib.require_gpu()
            key = (0, 0)
        return [key]
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be None, 'all' or a list of device indices")
        devices = list(range(_lib.device_count()))
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices is empty")
    count = _lib.device_count()
    lanes, seen = [], {}
    for d in devices:
        if d < 0 or d >= count:
            raise ValueError(f"no device {d}: {count} visible")
        bank = seen.get(d, 0)
        if bank >= 2:
            raise ValueError(f"device {d} named more than twice: the library keeps two scratch banks per device")
        seen[d] = bank + 1
        lanes.append((d, bank))
    return lanes[:max(1, n_chunks)]


def _frames(movie, c0: int, c1: int) -> np.ndarray:
    """Frames [c0, c1) of an ndarray / memmap (one slice) or of a movie object that only indexes by frame."""
    if isinstance(movie, np.ndarray):
        return movie[c0:c1]
    try:
        a = np.asarray(movie[c0:c1])
        if a.ndim == 3 and len(a) == c1 - c0:
            return a
    except Exception:
        pass
    return np.stack([np.asarray(movie[f]) for f in range(c0, c1)])


def localize_file(path: str, camera_info: dict, parameters: dict, *, fitting_method: str = "gaussmle", roi=None,
                  frame_bounds=None, eps: float = 0.001, max_it: int = 100, mle_method: str = "sigmaxy",
                  drift: int = 0, suffix: str = "_locs") -> str:
    """File to file, what `picasso localize movie.raw` does (picasso/__main__.py:1046-1156): read
    <name>.raw + .yaml, localize on the GPU, optionally RCC-undrift with `drift` frames per segment,
    write <name><suffix>.hdf5 + .yaml (and <name><suffix>_undrift.hdf5 + .yaml, <name><suffix>_drift.txt).  Returns the path written last."""
    import os

    from . import io, postprocess
    movie, info = io.load_movie(path)
    if drift and drift > 0 and len(movie):
        # the undrift below needs FFT plans for this frame size — seconds of kernel compilation inside rocFFT for large
        # frames: made on a side thread while the movie is localized (joined before the correlations, backend.py)
        backend.prewarm_fft(int(movie.shape[1]), int(movie.shape[2]))
    # the metadata the CLI saves: movie info + identify info + fit info (picasso/__main__.py:1086-1100, localize.py:1810)
    locs, info = localize(movie, camera_info, parameters, roi=roi, frame_bounds=frame_bounds, movie_info=info,
                          fitting_method=fitting_method, eps=eps, max_it=max_it, mle_method=mle_method, return_info=True)
    base = os.path.splitext(path)[0]
    out = base + suffix + ".hdf5"
    io.save_locs(out, locs, info)
    if drift and drift > 0:
        pinfo = info if lib.get_from_metadata(info, "Pixelsize") is not None else info + [{"Pixelsize": 130}]
        drift_table, locs = postprocess.undrift(locs, pinfo, drift, display=False)
        info = info + [{"Generated by": f"Picasso v{__version__} Undrift", "Segmentation": drift,       # __main__.py:443-481
                        "Drift X": float(drift_table["x"].mean()), "Drift Y": float(drift_table["y"].mean())}]
        out = base + suffix + "_undrift.hdf5"
        io.save_locs(out, locs, info)
        io.save_drift(base + suffix + "_drift.txt", drift_table)       # as `picasso undrift` does (__main__.py:476-482)
    return out


def _reference_module(given, name: str):
    if given is not None:
        return given
    try:
        import importlib
        return importlib.import_module("picasso." + name)
    except ImportError:
        return None


def install(picasso_localize=None, picasso_gaussmle=None, picasso_gausslq=None, picasso_zfit=None,
            picasso_render=None, picasso_imageprocess=None, picasso_postprocess=None, *, fused: bool = False,
            devices=None) -> None:
    """Rebind the reference package's hot-path functions to this backend, so that
    picasso.__main__ and the GUI run on the GPU unchanged.  Modules not given are taken
    from the installed ``picasso`` package; the rows next to the path (z fit, render, RCC undrift) are rebound
    when their module is available.  Rotated renders keep going to the reference's own functions.

    ``fused=True`` also rebinds ``picasso.localize.localize`` (what `picasso localize` calls, picasso/__main__.py:1086)
    to this package's: same arguments, table and metadata, but the movie crosses PCIe once (`localize_streamed`) instead
    of once for `identify` and once for `get_spots`.  ``devices`` (see `set_devices`) then spreads the frame chunks over
    several GPUs of the process."""
    if picasso_localize is None:
        import picasso.localize as picasso_localize       # the installed reference
    if picasso_gaussmle is None:
        import picasso.gaussmle as picasso_gaussmle
    import sys
    me = sys.modules[__name__]
    for name in ("identify", "identify_by_frame_number", "identify_in_frame", "identify_in_image", "get_spots"):
        setattr(picasso_localize, name, getattr(me, name))
    picasso_localize._fit2d_gaussmle = _fit2d_gaussmle
    picasso_localize._fit2d_gausslq = _fit2d_gausslq
    if fused:
        picasso_localize.localize = localize
    if devices is not None:
        set_devices(devices)
    for name in ("gaussmle", "gaussmle_async"):
        setattr(picasso_gaussmle, name, getattr(gaussmle, name))
    if picasso_gausslq is None:
        try:
            import picasso.gausslq as picasso_gausslq
        except ImportError:
            picasso_gausslq = None
    if picasso_gausslq is not None:
        for name in ("fit_spot", "fit_spots", "fit_spots_parallel", "fits_from_futures"):
            setattr(picasso_gausslq, name, getattr(gausslq, name))
    picasso_zfit = _reference_module(picasso_zfit, "zfit")
    if picasso_zfit is not None:
        from . import zfit as amd_zfit
        for name in ("_fit_z", "_fit_z_parallel", "locs_from_futures"):
            setattr(picasso_zfit, name, getattr(amd_zfit, name))
    picasso_render = _reference_module(picasso_render, "render")
    if picasso_render is not None:
        from . import render as amd_render

        def unrotated(mine, theirs):
            def call(locs, oversampling, y_min, x_min, y_max, x_max, *rest, **kw):
                n_blur = 1 if mine is not amd_render._render_hist else 0
                ang = kw.get("ang", rest[n_blur] if len(rest) > n_blur else None)
                if ang is not None and theirs is not None:
                    return theirs(locs, oversampling, y_min, x_min, y_max, x_max, *rest, **kw)
                return mine(locs, oversampling, y_min, x_min, y_max, x_max, *rest, **kw)
            call.__name__ = mine.__name__
            return call

        for name in ("_render_hist", "_render_gaussian", "_render_gaussian_iso"):
            setattr(picasso_render, name, unrotated(getattr(amd_render, name), getattr(picasso_render, name, None)))
    picasso_imageprocess = _reference_module(picasso_imageprocess, "imageprocess")
    if picasso_imageprocess is not None:
        from . import imageprocess as amd_ip
        for name in ("xcorr", "get_image_shift", "rcc"):
            setattr(picasso_imageprocess, name, getattr(amd_ip, name))
    picasso_postprocess = _reference_module(picasso_postprocess, "postprocess")
    if picasso_postprocess is not None:
        from . import postprocess as amd_pp
        for name in ("segment", "undrift"):
            setattr(picasso_postprocess, name, getattr(amd_pp, name))
