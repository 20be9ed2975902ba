"""picasso.imageprocess surface for RCC drift correction (picasso/imageprocess.py:27-217):
``xcorr``, ``get_image_shift``, ``rcc``.  The correlations, the centre crop, the peak search and
the fit window come from csrc/xcorr.hip (hipFFT, float64); the 5-parameter peak fit of 25 numbers
per pair is the reference's own scipy ``curve_fit`` call on the host.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import curve_fit

from . import backend, lib


def xcorr(imageA, imageB):
    """fftshift(real(ifft2(fft2(A) * conj(fft2(B))))) / sqrt(A.size)  (imageprocess.py:27-50)."""
    return backend.xcorr_array(imageA, imageB)


def _flat_2d_gaussian(coords, a, xc, yc, s, b):
    x, y = coords
    A = a * np.exp(-0.5 * ((x - xc) ** 2 + (y - yc) ** 2) / s**2) + b
    return A.flatten()


def _fit_peak(fit_roi, box, y_max_, x_max_, Y_, X_, Y, X):
    """imageprocess.py:109-159 from the fit window on: -> (-yc, -xc)."""
    fit_X = int(box / 2)
    y, x = np.mgrid[-fit_X:fit_X + 1, -fit_X:fit_X + 1]
    p0 = [fit_roi.max(), 0, 0, 1, fit_roi.min()]
    bounds = ([0, -np.inf, -np.inf, 0, 0], [np.inf, np.inf, np.inf, np.inf, np.inf])
    popt, _ = curve_fit(_flat_2d_gaussian, (x, y), fit_roi.flatten(), p0=p0, bounds=bounds)
    xc = popt[1] + X_ + x_max_
    yc = popt[2] + Y_ + y_max_
    xc -= np.floor(X / 2)
    yc -= np.floor(Y / 2)
    return -yc, -xc


def _shifts_of_pairs(segments, box, roi, pairs=None):
    """(-yc, -xc) of every pair (all i < j when `pairs` is None)."""
    segments = np.asarray(segments)
    _, Y, X = segments.shape
    peak, valid, rois, (Y_, X_) = backend.rcc_pairs_arrays(segments, roi, box, pairs)
    out = []
    for p in range(len(valid)):
        if valid[p] == 1:
            out.append(_fit_peak(rois[p], box, int(peak[p, 0]), int(peak[p, 1]), Y_, X_, Y, X))
        else:
            out.append((0, 0))          # empty image (imageprocess.py:85-86) or truncated fit window (:118-119)
    return out


def get_image_shift(imageA, imageB, box: int, roi: int | None = None, display: bool = False):
    """Shift from imageA to imageB (imageprocess.py:53-161).  ``display`` is a GUI aid of the
    reference and is ignored."""
    return _shifts_of_pairs(np.stack([np.asarray(imageA, np.float64), np.asarray(imageB, np.float64)]), box, roi)[0]


def rcc(segments, max_shift: float | None = None, callback=None):
    """Redundant cross-correlation (Wang et al. 2014; imageprocess.py:164-217): all pairwise shifts
    in one device call, then lib.minimize_shifts.  The callback sees 0 and then every pair index."""
    n_segments = len(segments)
    shifts_x = np.zeros((n_segments, n_segments))
    shifts_y = np.zeros((n_segments, n_segments))
    if callback is not None:
        callback(0)
    pairs = _shifts_of_pairs(np.asarray(segments, np.float64), 5, max_shift)
    flag = 0
    for i in range(n_segments - 1):
        for j in range(i + 1, n_segments):
            shifts_y[i, j], shifts_x[i, j] = pairs[flag]
            flag += 1
            if callback is not None:
                callback(flag)
    return lib.minimize_shifts(shifts_x, shifts_y)
