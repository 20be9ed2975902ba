"""picasso.zfit surface: astigmatic z from (sx, sy) on the HIP backend.

``zfit`` (picasso/zfit.py:465-579), ``_fit_z`` (:327-382), ``filter_z_fits``
(:674-703) and the axial precision (:805-922).  The bounded Brent minimisation per
localization runs in csrc/zfit.hip; the vectorised table math stays NumPy like the
reference's.  ``multiprocess`` is accepted and ignored (one device call).
"""
from __future__ import annotations

from typing import Callable, Literal

import numpy as np
import pandas as pd

from . import __version__, backend, gausslq, gaussmle, lib

_BATCH = 1 << 20


def _get_calib_size(coeffs, z):
    return (coeffs[0] * z**6 + coeffs[1] * z**5 + coeffs[2] * z**4 + coeffs[3] * z**3 + coeffs[4] * z**2
            + coeffs[5] * z + coeffs[6])


def _get_prime_calib_size(coeffs, z):
    return (6 * coeffs[0] * z**5 + 5 * coeffs[1] * z**4 + 4 * coeffs[2] * z**3 + 3 * coeffs[3] * z**2
            + 2 * coeffs[4] * z + coeffs[5])


def _sigma_standard_errors(locs: pd.DataFrame, fitting_method: str):
    """(se_sx, se_sy) in camera pixels: Mortensen-style estimate for the least-squares fit, the CRLB columns
    of the MLE table when present, else the Rieger-Stallinga estimate (picasso/zfit.py:844-871)."""
    if fitting_method == "gausslq":
        estimate = gausslq.sigma_uncertainty
    elif fitting_method == "gaussmle":
        if {"sx_unc", "sy_unc"} <= set(locs.columns):
            return locs["sx_unc"], locs["sy_unc"]
        estimate = gaussmle.sigma_uncertainty
    else:
        raise ValueError("fitting_method must be 'gausslq' or 'gaussmle'.")
    return (estimate(locs["sx"], locs["sy"], locs["photons"], locs["bg"]),
            estimate(locs["sy"], locs["sx"], locs["photons"], locs["bg"]))


def _axial_localization_precision_astig(locs: pd.DataFrame, cx, cy, magnification_factor: float, pixelsize: float,
                                        fitting_method: Literal["gausslq", "gaussmle"] = "gausslq"):
    """lpz in nm (picasso/zfit.py:805-889): error propagation through z = argmin of the calibration
    residual in sqrt(width).  Per axis a, with w_a(z) the calibrated width in nm:
        g_a = d sqrt(w_a) / dz = w_a'(z) / (2 sqrt(w_a(z))),   e_a = se(sigma_a) / (2 sqrt(sigma_a))  (nm units)
        lpz = sqrt(g_x^2 e_x^2 + g_y^2 e_y^2) / (g_x^2 + g_y^2)"""
    se = [v * pixelsize for v in _sigma_standard_errors(locs, fitting_method)]
    z = locs["z"] / magnification_factor       # the spot size actually measured
    slope_sq, err_sq = [], []
    for coeffs, width, se_axis in ((cx, locs["sx"], se[0]), (cy, locs["sy"], se[1])):
        w = _get_calib_size(coeffs, z) * pixelsize
        w_prime = _get_prime_calib_size(coeffs, z) * pixelsize
        slope_sq.append((w_prime / (2 * np.sqrt(w))) ** 2)
        err_sq.append(((1 / (2 * np.sqrt(width * pixelsize))) * se_axis) ** 2)
    lpz = np.sqrt((slope_sq[0] * err_sq[0] + slope_sq[1] * err_sq[1]) / (slope_sq[0] + slope_sq[1]) ** 2)
    return lpz * magnification_factor


def filter_z_fits(locs: pd.DataFrame, range: int) -> pd.DataFrame:
    """Keep fits whose calibration residual is at most `range` x the RMS residual (0 = keep all)."""
    if range > 0 and "d_zcalib" in locs.columns:
        residual = locs["d_zcalib"]
        locs = locs[residual <= range * np.sqrt(np.nanmean(residual ** 2))]
    return locs


def _fit_z(locs: pd.DataFrame, info, calibration: dict, magnification_factor: float, pixelsize: float,
           fitting_method: Literal["gausslq", "gaussmle"] = "gausslq", filter: int = 2,
           progress_callback=None, abort_callback=None):
    locs = locs.copy()
    cx = np.array(calibration["X Coefficients"])
    cy = np.array(calibration["Y Coefficients"])
    sx = locs["sx"].to_numpy()
    sy = locs["sy"].to_numpy()
    z = np.zeros_like(locs["x"])
    square_d_zcalib = np.zeros_like(z)
    N = len(z)
    bar = None
    if progress_callback == "console":
        from tqdm import tqdm
        bar = tqdm(total=N, desc="Fitting z...", unit="locs")
    for i0 in range(0, N, _BATCH):
        if callable(abort_callback) and abort_callback():
            return None
        i1 = min(N, i0 + _BATCH)
        zz, sq = backend.zfit_arrays(sx[i0:i1], sy[i0:i1], cx, cy)
        z[i0:i1] = zz
        square_d_zcalib[i0:i1] = sq
        if bar is not None:
            bar.update(i1 - i0)
        elif callable(progress_callback):
            progress_callback(i1 - 1)
    if bar is not None:
        bar.close()
    locs["z"] = z * magnification_factor
    locs["d_zcalib"] = np.sqrt(square_d_zcalib)
    locs["lpz"] = _axial_localization_precision_astig(locs, cx, cy, magnification_factor, pixelsize, fitting_method)
    locs = lib.ensure_sanity(locs, info)
    return filter_z_fits(locs, filter)


def fit_z(locs: pd.DataFrame, info, calibration: dict, magnification_factor: float, pixelsize: float,
          fitting_method: Literal["gausslq", "gaussmle"] = "gausslq", filter: int = 2, progress_callback=None):
    """The older public name of ``_fit_z`` (picasso/zfit.py:294-324): same result, with the reference's notice."""
    lib.deprecation_warning("Deprecation warning: `fit_z` will become a private function in v0.11.0. "
                            "Please use `zfit` instead.")
    return _fit_z(locs, info, calibration, magnification_factor, pixelsize, fitting_method, filter, progress_callback)


def _fit_z_parallel(locs: pd.DataFrame, info, calibration: dict, magnification_factor: float, pixelsize: float,
                    fitting_method: Literal["gausslq", "gaussmle"] = "gausslq", filter: int = 2, asynch: bool = False):
    """picasso/zfit.py:414-462 without the process pool: the device call is the parallel form.  With ``asynch`` the
    unfiltered result comes back as one finished future, which ``locs_from_futures`` filters like the reference."""
    if asynch:
        return [gausslq._DoneFuture(_fit_z(locs, info, calibration, magnification_factor, pixelsize,
                                           fitting_method=fitting_method, filter=0))]
    return locs_from_futures(_fit_z_parallel(locs, info, calibration, magnification_factor, pixelsize,
                                             fitting_method, filter, asynch=True), filter=filter)


def fit_z_parallel(locs: pd.DataFrame, info, calibration: dict, magnification_factor: float, pixelsize: float,
                   fitting_method: Literal["gausslq", "gaussmle"] = "gausslq", filter: int = 2, asynch: bool = False):
    """The older public name of ``_fit_z_parallel`` (picasso/zfit.py:385-411)."""
    lib.deprecation_warning("Deprecation warning: `fit_z_parallel` will become a private function in v0.11.0. "
                            "Please use `zfit` instead.")
    return _fit_z_parallel(locs, info, calibration, magnification_factor, pixelsize, fitting_method, filter, asynch)


def locs_from_futures(futures, filter: int = 2) -> pd.DataFrame:
    """Concatenate per-task z fits and apply the residual filter once (picasso/zfit.py:648-671)."""
    return filter_z_fits(pd.concat([f.result() for f in futures], ignore_index=True), filter)


def axial_localization_precision_astig(locs, info, calibration: dict,
                                       fitting_method: Literal["gausslq", "gaussmle"] = "gausslq"):
    """lpz (nm) of already z-fitted localizations from a calibration dictionary (picasso/zfit.py:747-803)."""
    assert fitting_method in ["gausslq", "gaussmle"], "fitting_method must be 'gausslq' or 'gaussmle'."
    assert ("X Coefficients" in calibration and "Y Coefficients" in calibration
            and "Magnification factor" in calibration), (
        "Calibration dictionary must contain 'X Coefficients', 'Y Coefficients', and 'Magnification factor'.")
    pixelsize = lib.get_from_metadata(info, "Pixelsize")
    if pixelsize is None:
        raise ValueError("Pixelsize not found in info.")
    return _axial_localization_precision_astig(locs, np.array(calibration["X Coefficients"]),
                                               np.array(calibration["Y Coefficients"]),
                                               calibration["Magnification factor"], pixelsize, fitting_method)


def axial_localization_precision(locs, info, calibration: dict,
                                 fitting_method: Literal["gausslq", "gaussmle"] = "gausslq",
                                 modality: Literal["astigmatic"] = "astigmatic"):
    """picasso/zfit.py:706-744: dispatch on the 3D modality (astigmatism is the only one)."""
    if modality != "astigmatic":
        raise NotImplementedError("Currently only 'astigmatic' modality is supported.")
    return axial_localization_precision_astig(locs, info, calibration, fitting_method)


def zfit(locs: pd.DataFrame, info, *, calibration: dict, magnification_factor: float | None = None,
         pixelsize: int | float | None = None, fitting_method: Literal["gausslq", "gaussmle"] = "gausslq",
         filter: int = 2, multiprocess: bool = False,
         progress_callback: Callable[[int], None] | Literal["console"] | None = None,
         abort_callback: Callable[[], bool] | None = None):
    """Fit z to 2D-fitted localizations; returns (locs, info) or (None, None) when aborted."""
    assert fitting_method in ["gausslq", "gaussmle"], "Invalid fitting method."
    assert filter >= 0, "Filter must be non-negative."
    assert isinstance(calibration, dict), "Calibration must be a dict, see ``io.load_calibration``."
    if magnification_factor is not None:
        assert isinstance(magnification_factor, (int, float)), "Magnification factor must be a number."
        calibration["Magnification factor"] = float(magnification_factor)
    else:
        assert "Magnification factor" in calibration, "Magnification factor is missing in calibration."
    if pixelsize is not None:
        assert isinstance(pixelsize, (int, float)), "Pixelsize must be a number in nm."
        pixelsize = float(pixelsize)
        info.append({"Pixelsize": pixelsize})
    else:
        assert lib.get_from_metadata(info, "Pixelsize") is not None, (
            "Camera pixel size (nm) is missing. Enter it either in the info metadata, or as an argument.")
    pixelsize = lib.get_from_metadata(info, "Pixelsize", raise_error=True)
    out = _fit_z(locs, info, calibration, calibration["Magnification factor"], pixelsize, fitting_method, filter,
                 progress_callback, abort_callback)
    if out is None:
        return None, None
    new_info = {"Generated by": f"Picasso v{__version__} Fit 3D",
                "Calibration path": calibration.get("Path", "N/A"), "Filter range": filter}
    return out, info + [new_info | calibration]
