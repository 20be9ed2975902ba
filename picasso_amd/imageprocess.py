"""picasso.imageprocess surface for RCC drift correction (picasso/imageprocess.py:27-217):
``xcorr``, ``get_image_shift``, ``rcc``.  The correlations, the centre crop, the peak search, the fit window
(csrc/xcorr.hip, hipFFT, float64) and the bounded Gaussian fit of the window — the reference's scipy
``curve_fit`` call, restated as a Trust Region Reflective solver with one thread per pair (csrc/peakfit.hip) —
all run on the device.
"""
from __future__ import annotations

import numpy as np

from . import backend, lib


def xcorr(imageA, imageB):
    """fftshift(real(ifft2(fft2(A) * conj(fft2(B))))) / sqrt(A.size)  (imageprocess.py:27-50)."""
    return backend.xcorr_array(imageA, imageB)


def _shifts_of_pairs(segments, box, roi, pairs=None):
    """(-yc, -xc) of every pair (all i < j when `pairs` is None).  Pairs with an empty image (imageprocess.py:85-86)
    or a fit window truncated by the border (:118-119) give (0, 0), as in the reference."""
    segments = np.asarray(segments)
    shifts, status = backend.rcc_shifts_arrays(segments, roi, box, pairs)
    # what scipy.optimize.curve_fit raises at picasso/imageprocess.py:129-135, in its order: non-finite input, a start
    # value outside the bounds (b = window minimum < 0), no convergence within max_nfev
    if np.any(status == -3):
        raise ValueError("array must not contain infs or NaNs")
    if np.any(status == -2):
        raise ValueError("Initial guess is outside of provided bounds")
    if np.any(status == 0):
        raise RuntimeError("Optimal parameters not found: The maximum number of function evaluations is exceeded.")
    return [(float(sy), float(sx)) if st != -1 else (0, 0) for (sy, sx), st in zip(shifts, status)]


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
