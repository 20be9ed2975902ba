"""picasso.render surface for the two render modes on the drift-correction / display path:
``blur_method=None`` (2-D histogram), ``"gaussian"`` (one separable Gaussian per localization,
widths = localization precisions) and ``"gaussian_iso"`` (one width, their mean).  picasso/render.py:37-175 ``render``,
:798-853 ``_render_hist``, :1020-1070 ``_render_gaussian``; the pixels are computed by
csrc/render.hip.  Rotated views (``ang``) and the other blur methods are not built; they
raise instead of falling back to a CPU path.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import backend, lib

_NOT_BUILT = ("smooth", "convolve")


def _viewport(info, viewport):
    if viewport is None:
        try:
            viewport = [(0, 0), (info[0]["Height"], info[0]["Width"])]
        except TypeError:
            raise ValueError("Need info if no viewport is provided.")
    return viewport


def render(locs: pd.DataFrame, info, oversampling: float = 1.0, viewport=None, blur_method=None,
           min_blur_width: float = 0.0, ang=None, disp_px_size: float | None = None):
    """-> (n, image): number of localizations rendered and the float32 image."""
    pixelsize = lib.get_from_metadata(info, "Pixelsize", raise_error=True)
    if disp_px_size is None:
        lib.deprecation_warning("Deprecation warning: the 'oversampling' parameter is deprecated and will be removed "
                                "in v0.11.0. Use 'disp_px_size' instead.")
        disp_px_size = pixelsize / oversampling
    oversampling = pixelsize / disp_px_size
    (y_min, x_min), (y_max, x_max) = _viewport(info, viewport)
    if ang is not None:
        raise NotImplementedError("rotated rendering (ang) has no HIP kernel in picasso_amd")
    if blur_method is None:
        return _render_hist(locs, oversampling, y_min, x_min, y_max, x_max)
    if blur_method == "gaussian":
        return _render_gaussian(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width)
    if blur_method == "gaussian_iso":
        return _render_gaussian_iso(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width)
    if blur_method in _NOT_BUILT:
        raise NotImplementedError(f"blur_method={blur_method!r} has no HIP kernel in picasso_amd; there is no CPU fallback")
    raise Exception("blur_method not understood.")


def _render_hist(locs, oversampling, y_min, x_min, y_max, x_max, ang=None):
    if ang is not None:
        raise NotImplementedError("rotated rendering (ang) has no HIP kernel in picasso_amd")
    return backend.render_arrays(locs["x"].to_numpy(), locs["y"].to_numpy(), oversampling, y_min, x_min, y_max, x_max)


def _render_gaussian(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=None):
    if ang is not None:
        raise NotImplementedError("rotated rendering (ang) has no HIP kernel in picasso_amd")
    return backend.render_arrays(locs["x"].to_numpy(), locs["y"].to_numpy(), oversampling, y_min, x_min, y_max, x_max,
                                 lpx=locs["lpx"].to_numpy(), lpy=locs["lpy"].to_numpy(), min_blur_width=min_blur_width)


def _render_gaussian_iso(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=None):
    """picasso/render.py:1148-1216: one isotropic width per localization, the mean of the two."""
    if ang is not None:
        raise NotImplementedError("rotated rendering (ang) has no HIP kernel in picasso_amd")
    return backend.render_arrays(locs["x"].to_numpy(), locs["y"].to_numpy(), oversampling, y_min, x_min, y_max, x_max,
                                 lpx=locs["lpx"].to_numpy(), lpy=locs["lpy"].to_numpy(), min_blur_width=min_blur_width,
                                 iso=True)


def _older_name(name: str) -> None:
    lib.deprecation_warning(f"Deprecation warning: the '{name}' function is deprecated and will be removed in "
                            f"v0.11.0. Use _{name} instead if necessary.")


def render_hist(locs, oversampling, y_min, x_min, y_max, x_max, ang=None):
    """Older public name of ``_render_hist`` (picasso/render.py:776-795)."""
    _older_name("render_hist")
    return _render_hist(locs, oversampling, y_min, x_min, y_max, x_max, ang=ang)


def render_gaussian(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=None):
    """Older public name of ``_render_gaussian`` (picasso/render.py:990-1017) — the call BASELINE.json's
    config 3 is quoted on."""
    _older_name("render_gaussian")
    return _render_gaussian(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=ang)


def render_gaussian_iso(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=None):
    """Older public name of ``_render_gaussian_iso`` (picasso/render.py:1118-1145)."""
    _older_name("render_gaussian_iso")
    return _render_gaussian_iso(locs, oversampling, y_min, x_min, y_max, x_max, min_blur_width, ang=ang)
