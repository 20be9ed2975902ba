"""ctypes front-end of the CPU oracle (oracle/picasso_oracle.c).

TEST INFRASTRUCTURE.  Import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (picasso_amd) must never
import this module.

Every function mirrors a reference function; see the C file for file:line.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpicasso_oracle.so")
_lib = None

DTYPE_CODES = {
    np.dtype("uint16"): 0, np.dtype("uint8"): 1, np.dtype("int16"): 2,
    np.dtype("uint32"): 3, np.dtype("int32"): 4, np.dtype("float32"): 5,
}
METHODS = {"sigma": 0, "sigmaxy": 1}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "picasso_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, f64, i32 = ctypes.c_int64, ctypes.c_double, ctypes.c_int
        p = ctypes.c_void_p
        L.orc_identify.argtypes = [p, i32, i64, i64, i64, i32, f64, p, i64, i64,
                                   p, p, p, p, i64, p, i32]
        L.orc_identify.restype = i32
        L.orc_get_spots.argtypes = [p, i32, i64, i64, i64, p, p, p, i64, i32, f64, f64, f64, p]
        L.orc_get_spots.restype = i32
        L.orc_gaussmle.argtypes = [p, i64, i32, f64, i32, i32, p, p, p, p, i32]
        L.orc_gaussmle.restype = i32
        L.orc_gaussmle_closeness.argtypes = [p, i64, i32, f64, i32, i32, p, p, p, p, p, i32]
        L.orc_gaussmle_closeness.restype = i32
        L.orc_peak_fit.argtypes = [p, i32, p, p]
        L.orc_peak_fit.restype = i32
        L.orc_initial_parameters.argtypes = [p, i64, i32, p]
        L.orc_initial_parameters.restype = i32
        L.orc_unit_vectors.argtypes = [i32, p, p]
        L.orc_avgroi.argtypes = [p, i64, i32, p]
        L.orc_gausslq.argtypes = [p, i64, i32, p, p, p, i32]
        L.orc_gausslq_initial.argtypes = [p, i64, i32, p]
        L.orc_gausslq_from.argtypes = [p, i64, i32, p, p, i32]
        L.orc_zfit.argtypes = [p, p, i64, p, p, p, p, i32]
        L.orc_net_gradient.argtypes = [p, i64, i64, p, p, i64, i32, p, p, p]
        L.orc_net_gradient.restype = i32
        L.orc_max_threads.restype = i32
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads() -> int:
    return int(lib().orc_max_threads())


def normalise_roi(roi, Y, X):
    """numpy slice semantics of ``frame[y0:y1, x0:x1]`` (localize.py:331)."""
    if roi is None:
        return None
    (y0, x0), (y1, x1) = roi
    ys, ye, _ = slice(y0, y1).indices(Y)
    xs, xe, _ = slice(x0, x1).indices(X)
    return np.array([ys, xs, max(ye, ys), max(xe, xs)], np.int64)


def frame_range(frame_bounds, F):
    """Inclusive frame range of localize.py:395-401."""
    lo, hi = 0, F
    if frame_bounds is not None:
        if frame_bounds[0] is not None:
            lo = max(frame_bounds[0], lo)
        if frame_bounds[1] is not None:
            hi = min(frame_bounds[1], hi)
    return lo, hi


def identify(movie, min_ng, box, roi=None, frame_bounds=None, threads=1):
    """-> frame, y, x (int64) and net_gradient (float32), frame/y/x ordered."""
    movie = np.ascontiguousarray(movie)
    F, Y, X = movie.shape
    code = DTYPE_CODES[movie.dtype]
    r = normalise_roi(roi, Y, X)
    lo, hi = frame_range(frame_bounds, F)
    cap = max(1024, F * 512)
    while True:
        fr = np.empty(cap, np.int64); yy = np.empty(cap, np.int64)
        xx = np.empty(cap, np.int64); ng = np.empty(cap, np.float32)
        n = ctypes.c_int64(0)
        rc = lib().orc_identify(_ptr(movie), code, F, Y, X, int(box), float(min_ng),
                                _ptr(r) if r is not None else None, lo, hi,
                                _ptr(fr), _ptr(yy), _ptr(xx), _ptr(ng), cap,
                                ctypes.byref(n), int(threads))
        if rc < 0:
            raise ValueError(f"orc_identify failed ({rc})")
        if rc == 0:
            k = n.value
            return fr[:k].copy(), yy[:k].copy(), xx[:k].copy(), ng[:k].copy()
        cap = n.value


def get_spots(movie, frame, y, x, box, camera_info):
    movie = np.ascontiguousarray(movie)
    F, Y, X = movie.shape
    frame = np.ascontiguousarray(frame, np.int64)
    y = np.ascontiguousarray(y, np.int64)
    x = np.ascontiguousarray(x, np.int64)
    N = len(frame)
    spots = np.empty((N, box, box), np.float32)
    lib().orc_get_spots(_ptr(movie), DTYPE_CODES[movie.dtype], F, Y, X, _ptr(frame), _ptr(y),
                        _ptr(x), N, int(box), float(camera_info["Baseline"]),
                        float(camera_info["Sensitivity"]), float(camera_info["Gain"]), _ptr(spots))
    return spots


def gaussmle(spots, eps, max_it, method="sigmaxy", threads=1):
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    if method not in METHODS:
        raise ValueError("Method not available.")
    thetas = np.zeros((N, 6), np.float32)
    crlbs = np.full((N, 6), np.inf, np.float32)
    ll = np.zeros(N, np.float32)
    it = np.zeros(N, np.int32)
    rc = lib().orc_gaussmle(_ptr(spots), N, box, float(eps), int(max_it), METHODS[method],
                            _ptr(thetas), _ptr(crlbs), _ptr(ll), _ptr(it), int(threads))
    if rc != 0:
        raise ValueError(f"orc_gaussmle failed ({rc})")
    return thetas, crlbs, ll, it


def gaussmle_closeness(spots, eps, max_it, method="sigmaxy", threads=1):
    """gaussmle plus, per spot, min over the iterations of |D / eps - 1| (D = the largest step the convergence
    test looks at): the relative margin by which the fit's stop / continue decisions were taken."""
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    if method not in METHODS:
        raise ValueError("Method not available.")
    thetas = np.zeros((N, 6), np.float32)
    crlbs = np.full((N, 6), np.inf, np.float32)
    ll = np.zeros(N, np.float32)
    it = np.zeros(N, np.int32)
    close = np.zeros(N, np.float32)
    rc = lib().orc_gaussmle_closeness(_ptr(spots), N, box, float(eps), int(max_it), METHODS[method],
                                      _ptr(thetas), _ptr(crlbs), _ptr(ll), _ptr(it), _ptr(close), int(threads))
    if rc != 0:
        raise ValueError(f"orc_gaussmle_closeness failed ({rc})")
    return thetas, crlbs, ll, it, close


def peak_fit(roi):
    """Bounded Gaussian fit of a (box, box) float64 correlation window, the reference's curve_fit call
    (picasso/imageprocess.py:121-141) -> popt (a, xc, yc, s, b), scipy's termination status, nfev."""
    roi = np.ascontiguousarray(roi, np.float64)
    if roi.min() < 0:
        raise ValueError("Initial guess is outside of provided bounds")      # what curve_fit raises for b0 < 0
    popt = np.zeros(5, np.float64)
    nfev = ctypes.c_int(0)
    status = lib().orc_peak_fit(_ptr(roi), int(roi.shape[0]), _ptr(popt), ctypes.byref(nfev))
    return popt, int(status), int(nfev.value)


def image_shift_from_window(win, box, y_max_, x_max_, Y_, X_, Y, X):
    """picasso/imageprocess.py:109-159 from the fit window on: -> (-yc, -xc)."""
    popt, _, _ = peak_fit(win)
    xc = popt[1] + X_ + x_max_
    yc = popt[2] + Y_ + y_max_
    xc -= np.floor(X / 2)
    yc -= np.floor(Y / 2)
    return -yc, -xc


def initial_parameters(spots):
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    th = np.zeros((N, 6), np.float32)
    lib().orc_initial_parameters(_ptr(spots), N, box, _ptr(th))
    return th


def unit_vectors(box):
    ux = np.zeros((box, box), np.float32); uy = np.zeros((box, box), np.float32)
    lib().orc_unit_vectors(int(box), _ptr(ux), _ptr(uy))
    return ux, uy


def net_gradient(frame, y, x, box, uy, ux):
    """picasso/localize.py:202-244 on one float32 frame -> float32 (len(y),)."""
    img = np.ascontiguousarray(frame, np.float32)
    y = np.ascontiguousarray(y, np.int32); x = np.ascontiguousarray(x, np.int32)
    uy = np.ascontiguousarray(uy, np.float32); ux = np.ascontiguousarray(ux, np.float32)
    out = np.zeros(len(y), np.float32)
    rc = lib().orc_net_gradient(_ptr(img), img.shape[0], img.shape[1], _ptr(y), _ptr(x), len(y), int(box),
                                _ptr(uy), _ptr(ux), _ptr(out))
    if rc:
        raise ValueError("net_gradient: a window reaches past the far edge of the image")
    return out


def avgroi(spots):
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    th = np.zeros((N, 6), np.float32)
    lib().orc_avgroi(_ptr(spots), N, box, _ptr(th))
    return th


def zfit(sx, sy, cx, cy, threads=1):
    """Bounded Brent per localization -> z (un-magnified) and squared residual, float64."""
    sx = np.ascontiguousarray(sx, np.float32); sy = np.ascontiguousarray(sy, np.float32)
    cx = np.ascontiguousarray(cx, np.float64); cy = np.ascontiguousarray(cy, np.float64)
    N = len(sx)
    z = np.zeros(N, np.float64); sq = np.zeros(N, np.float64)
    lib().orc_zfit(_ptr(sx), _ptr(sy), N, _ptr(cx), _ptr(cy), _ptr(z), _ptr(sq), int(threads))
    return z, sq


def gausslq(spots, threads=1, full=False):
    """gausslq.fit_spots restated (MINPACK lmdif): theta (N,6) float32 [x, y, photons, bg, sx, sy]."""
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    th = np.full((N, 6), np.nan, np.float32)
    info = np.zeros(N, np.int32); nfev = np.zeros(N, np.int32)
    lib().orc_gausslq(_ptr(spots), N, box, _ptr(th), _ptr(info), _ptr(nfev), int(threads))
    return (th, info, nfev) if full else th


def lq_set_exp(which: int):
    """PROBE: 0 = libm's exp (default), 1 = exp correctly rounded to float64 (picasso_oracle.c orc_lq_set_exp)."""
    lib().orc_lq_set_exp(int(which))


def gausslq_initial(spots):
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    th = np.zeros((N, 6), np.float32)
    lib().orc_gausslq_initial(_ptr(spots), N, box, _ptr(th))
    return th


def gausslq_from(spots, theta0, threads=1):
    spots = np.ascontiguousarray(spots, np.float32)
    theta0 = np.ascontiguousarray(theta0, np.float32)
    N, box, _ = spots.shape
    th = np.zeros((N, 6), np.float32)
    lib().orc_gausslq_from(_ptr(spots), N, box, _ptr(theta0), _ptr(th), int(threads))
    return th


def render(x, y, oversampling, viewport, lpx=None, lpy=None, blur_method=None, min_blur_width=0.0):
    """render._render_hist / _render_gaussian restated: -> (n, image float32)."""
    import ctypes
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    (y_min, x_min), (y_max, x_max) = viewport
    ny = ctypes.c_int64(); nx = ctypes.c_int64()
    L = lib()
    f64 = ctypes.c_double
    L.orc_render_dims.argtypes = [f64] * 5 + [ctypes.c_void_p] * 2
    L.orc_render_dims(float(oversampling), float(y_min), float(x_min), float(y_max), float(x_max), ctypes.byref(ny), ctypes.byref(nx))
    image = np.zeros((ny.value, nx.value), np.float32)
    if blur_method is None:
        L.orc_render_hist.restype = ctypes.c_int64
        L.orc_render_hist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + [f64] * 5 + [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        n = L.orc_render_hist(_ptr(x), _ptr(y), len(x), float(oversampling), float(y_min), float(x_min), float(y_max),
                              float(x_max), _ptr(image), ny.value, nx.value)
    elif blur_method in ("gaussian", "gaussian_iso"):
        lpx = np.ascontiguousarray(lpx, np.float32); lpy = np.ascontiguousarray(lpy, np.float32)
        L.orc_render_gaussian.restype = ctypes.c_int64
        L.orc_render_gaussian.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [f64] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        n = L.orc_render_gaussian(_ptr(x), _ptr(y), _ptr(lpx), _ptr(lpy), len(x), float(oversampling), float(y_min),
                                  float(x_min), float(y_max), float(x_max), float(min_blur_width),
                                  int(blur_method == "gaussian_iso"), _ptr(image), ny.value, nx.value)
        if n < 0:
            raise ValueError("footprint too large for the oracle")
    else:
        raise ValueError("blur_method not understood.")
    return int(n), image


def xcorr(image_a, image_b):
    """imageprocess.xcorr restated with numpy's FFT (picasso/imageprocess.py:27-50)."""
    fa = np.fft.fft2(image_a)
    cfb = np.conj(np.fft.fft2(image_b))
    return np.fft.fftshift(np.real(np.fft.ifft2(fa * cfb))) / np.sqrt(np.asarray(image_a).size)


def peak_window(image_a, image_b, box, roi=None):
    """get_image_shift up to its curve_fit (picasso/imageprocess.py:85-119):
    -> None for an empty image, else (y_max, x_max, Y_, X_, fit window or None if truncated)."""
    if np.sum(image_a) == 0 or np.sum(image_b) == 0:
        return None
    xc = xcorr(image_a, image_b)
    Y, X = np.asarray(image_a).shape
    if roi is not None:
        Y_ = int((Y - roi) / 2)
        X_ = int((X - roi) / 2)
        if Y_ > 0:
            xc = xc[Y_:-Y_, :]
        else:
            Y_ = 0
        if X_ > 0:
            xc = xc[:, X_:-X_]
        else:
            X_ = 0
    else:
        Y_ = X_ = 0
    h = int(box / 2)
    y_max_, x_max_ = np.unravel_index(xc.argmax(), xc.shape)
    win = xc[y_max_ - h:y_max_ + h + 1, x_max_ - h:x_max_ + h + 1] if (y_max_ - h >= 0 and x_max_ - h >= 0) else np.zeros((0, 0))
    if 0 in win.shape or win.shape[0] != win.shape[1] or win.shape[0] != box:
        win = None
    return int(y_max_), int(x_max_), Y_, X_, win
