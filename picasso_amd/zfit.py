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


def _axial_localization_precision_astig(locs: pd.DataFrame, cx, cy, magnification_factor: float, pixelsize: float,
                                        fitting_method: Literal["gausslq", "gaussmle"] = "gausslq"):
    """lpz in nm (picasso/zfit.py:805-889)."""
    if fitting_method == "gausslq":
        se_sx = gausslq.sigma_uncertainty(locs["sx"], locs["sy"], locs["photons"], locs["bg"]) * pixelsize
        se_sy = gausslq.sigma_uncertainty(locs["sy"], locs["sx"], locs["photons"], locs["bg"]) * pixelsize
    elif fitting_method == "gaussmle":
        if "sx_unc" not in locs.columns or "sy_unc" not in locs.columns:
            se_sx = gaussmle.sigma_uncertainty(locs["sx"], locs["sy"], locs["photons"], locs["bg"]) * pixelsize
            se_sy = gaussmle.sigma_uncertainty(locs["sy"], locs["sx"], locs["photons"], locs["bg"]) * pixelsize
        else:
            se_sx = locs["sx_unc"] * pixelsize
            se_sy = locs["sy_unc"] * pixelsize
    else:
        raise ValueError("fitting_method must be 'gausslq' or 'gaussmle'.")
    z = locs["z"] / magnification_factor       # the spot size actually measured
    wx_calib = _get_calib_size(cx, z) * pixelsize
    wy_calib = _get_calib_size(cy, z) * pixelsize
    wx_calib_prime = _get_prime_calib_size(cx, z) * pixelsize
    wy_calib_prime = _get_prime_calib_size(cy, z) * pixelsize
    sqrt_wx_calib = np.sqrt(wx_calib)
    sqrt_wx_calib_prime = wx_calib_prime / (2 * sqrt_wx_calib)
    sqrt_wy_calib = np.sqrt(wy_calib)
    sqrt_wy_calib_prime = wy_calib_prime / (2 * sqrt_wy_calib)
    delta_sqrt_wx = (1 / (2 * np.sqrt(locs["sx"] * pixelsize))) * se_sx
    delta_sqrt_wy = (1 / (2 * np.sqrt(locs["sy"] * pixelsize))) * se_sy
    swxc2 = sqrt_wx_calib_prime**2
    swyc2 = sqrt_wy_calib_prime**2
    swx2 = delta_sqrt_wx**2
    swy2 = delta_sqrt_wy**2
    lpz = np.sqrt((swxc2 * swx2 + swyc2 * swy2) / (swxc2 + swyc2) ** 2)
    return lpz * magnification_factor


def filter_z_fits(locs: pd.DataFrame, range: int) -> pd.DataFrame:
    """Drop fits whose calibration residual exceeds `range` x the RMS residual."""
    if "d_zcalib" not in locs.columns:
        return locs
    if range > 0:
        rmsd = np.sqrt(np.nanmean(locs["d_zcalib"] ** 2))
        locs = locs[locs["d_zcalib"] <= range * rmsd]
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
    new_info = {"Generated by": f"Picasso v{__version__} Fit 3D (picasso_amd HIP backend)",
                "Calibration path": calibration.get("Path", "N/A"), "Filter range": filter}
    return out, info + [new_info | calibration]
