"""Array-level calls into libpicasso_hip.so (numpy in, numpy out).

This is the thin layer the Picasso-shaped modules (localize.py, gaussmle.py)
sit on.  Nothing here computes on the CPU: every function ends in a C-ABI call
and raises if the library or the GPU is missing.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

LQ_COLUMNS = [
    ("frame", np.uint32), ("x", np.float32), ("y", np.float32), ("photons", np.float32),
    ("sx", np.float32), ("sy", np.float32), ("bg", np.float32), ("lpx", np.float32),
    ("lpy", np.float32), ("ellipticity", np.float32), ("net_gradient", np.float32),
]
LOC_COLUMNS = [
    ("frame", np.uint32), ("x", np.float32), ("y", np.float32), ("photons", np.float32),
    ("sx", np.float32), ("sy", np.float32), ("bg", np.float32), ("lpx", np.float32),
    ("lpy", np.float32), ("ellipticity", np.float32), ("net_gradient", np.float32),
    ("log_likelihood", np.float32), ("iterations", np.uint32), ("photons_unc", np.float32),
    ("bg_unc", np.float32), ("sx_unc", np.float32), ("sy_unc", np.float32),
]


def dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt not in _lib.DTYPE_CODES:
        raise TypeError(f"unsupported movie dtype {dt}; supported: "
                        + ", ".join(str(d) for d in _lib.DTYPE_CODES))
    return _lib.DTYPE_CODES[dt]


def as_movie_array(movie) -> np.ndarray:
    """C-contiguous native-endian (F, Y, X) view/copy of an ndarray or memmap."""
    a = np.asarray(movie)
    if a.ndim != 3:
        raise ValueError("movie must have shape (frames, height, width)")
    if not a.dtype.isnative:
        a = a.astype(a.dtype.newbyteorder("="))
    dtype_code(a.dtype)
    return np.ascontiguousarray(a)


def normalise_roi(roi, Y, X):
    """numpy slice semantics of frame[y0:y1, x0:x1] (picasso/localize.py:331)."""
    if roi is None:
        return None
    (y0, x0), (y1, x1) = roi
    ys, ye, _ = slice(y0, y1).indices(Y)
    xs, xe, _ = slice(x0, x1).indices(X)
    return np.array([ys, xs, max(ye, ys), max(xe, xs)], np.int64)


def frame_range(frame_bounds, F):
    """Inclusive range of picasso/localize.py:395-401 -> (lo, hi)."""
    lo, hi = 0, F
    if frame_bounds is not None:
        if frame_bounds[0] is not None:
            lo = max(frame_bounds[0], lo)
        if frame_bounds[1] is not None:
            hi = min(frame_bounds[1], hi)
    return int(lo), int(hi)


def identify_arrays(movie: np.ndarray, min_ng: float, box: int, roi=None, frame_bounds=None,
                    f_lo=None, f_hi=None):
    """-> frame, y, x (int32), net_gradient (float32), ordered by (frame, y, x)."""
    _lib.require_gpu()
    movie = as_movie_array(movie)
    F, Y, X = movie.shape
    r = normalise_roi(roi, Y, X)
    lo, hi = frame_range(frame_bounds, F)
    if f_lo is not None:
        lo = max(lo, f_lo)
    if f_hi is not None:
        hi = min(hi, f_hi)
    L = _lib.load()
    cap = max(4096, 256 * F)
    while True:
        fr = np.empty(cap, np.int32); yy = np.empty(cap, np.int32)
        xx = np.empty(cap, np.int32); ng = np.empty(cap, np.float32)
        n = ctypes.c_int64(0)
        with _lib.lock():
            rc = L.pmi_identify(_lib.ptr(movie), dtype_code(movie.dtype), F, Y, X, int(box), float(min_ng),
                                _lib.ptr(r), lo, hi, _lib.ptr(fr), _lib.ptr(yy), _lib.ptr(xx), _lib.ptr(ng),
                                cap, ctypes.byref(n))
        if rc == _lib.PMI_ERR_CAPACITY:
            cap = int(n.value)
            continue
        _lib.check(rc, "pmi_identify")
        k = int(n.value)
        return fr[:k].copy(), yy[:k].copy(), xx[:k].copy(), ng[:k].copy()


def net_gradient_array(image, y, x, box: int, uy, ux) -> np.ndarray:
    """float32 net gradient at the given pixels of one image (pmi_net_gradient)."""
    _lib.require_gpu()
    img = np.ascontiguousarray(image, np.float32)
    if img.ndim != 2:
        raise ValueError("image must be 2-D")
    y = np.ascontiguousarray(y, np.int32)
    x = np.ascontiguousarray(x, np.int32)
    uy = np.ascontiguousarray(uy, np.float32)
    ux = np.ascontiguousarray(ux, np.float32)
    if uy.shape != (box, box) or ux.shape != (box, box) or len(y) != len(x):
        raise ValueError("uy, ux must have shape (box, box) and y, x the same length")
    out = np.zeros(len(y), np.float32)
    with _lib.lock():
        rc = _lib.load().pmi_net_gradient(_lib.ptr(img), img.shape[0], img.shape[1], _lib.ptr(y), _lib.ptr(x), len(y),
                                          int(box), _lib.ptr(uy), _lib.ptr(ux), _lib.ptr(out))
    _lib.check(rc, "pmi_net_gradient")
    return out


def get_spots_array(movie: np.ndarray, frame, y, x, box: int, baseline, sensitivity, gain) -> np.ndarray:
    _lib.require_gpu()
    movie = as_movie_array(movie)
    F, Y, X = movie.shape
    frame = np.ascontiguousarray(frame, np.int32)
    y = np.ascontiguousarray(y, np.int32)
    x = np.ascontiguousarray(x, np.int32)
    N = len(frame)
    spots = np.empty((N, box, box), np.float32)
    with _lib.lock():
        rc = _lib.load().pmi_get_spots(_lib.ptr(movie), dtype_code(movie.dtype), F, Y, X, _lib.ptr(frame),
                                       _lib.ptr(y), _lib.ptr(x), N, int(box), float(baseline),
                                       float(sensitivity), float(gain), _lib.ptr(spots))
    _lib.check(rc, "pmi_get_spots")
    return spots


MLE_MODES = {"fast": 0, "refit": 1, "strict": 2}


def set_mle_mode(mode: str = "refit", margin: float = 0.001):
    """How the Newton loop of the MLE fit runs (pmi_mle_set_mode): "fast" = float32 loop only; "refit" (default) =
    float32 loop, then spots whose convergence decision (picasso/gaussmle.py:844-852) came within `margin` of eps
    are fitted again in the reference's float64-intermediate arithmetic; "strict" = every spot that way."""
    if mode not in MLE_MODES:
        raise ValueError(f"unknown MLE mode {mode!r}")
    _lib.check(_lib.load().pmi_mle_set_mode(MLE_MODES[mode], float(margin)), "pmi_mle_set_mode")


def get_mle_mode():
    m, g = ctypes.c_int(0), ctypes.c_double(0)
    _lib.check(_lib.load().pmi_mle_get_mode(ctypes.byref(m), ctypes.byref(g)), "pmi_mle_get_mode")
    return {v: k for k, v in MLE_MODES.items()}[m.value], g.value


MLE_LIBMS = {"device": 0, "glibc": 1, "auto": 2}


def set_mle_libm(which: str = "auto"):
    """Whose erf / exp the reference-arithmetic MLE kernel evaluates (pmi_mle_set_libm): "glibc" = the bits of the C library
    the reference's math.erf / math.exp resolve to under numba (picasso/gaussmle.py:279, 295), 12 - 18 % slower; "device" = the
    device library's functions; "auto" (default) = glibc's for every spot of the strict mode and for the re-fit of boxes up
    to 5x5, the device library's in the re-fit of larger boxes."""
    if which not in MLE_LIBMS:
        raise ValueError(f"unknown libm {which!r}")
    _lib.check(_lib.load().pmi_mle_set_libm(MLE_LIBMS[which]), "pmi_mle_set_libm")


def get_mle_libm() -> str:
    w = ctypes.c_int(0)
    _lib.check(_lib.load().pmi_mle_get_libm(ctypes.byref(w)), "pmi_mle_get_libm")
    return {v: k for k, v in MLE_LIBMS.items()}[w.value]


def last_refit_count(stream=None) -> int:
    """Spots the last MLE call OF THE CALLING THREAD fitted a second time (synchronises `stream`).  The library keeps these
    statistics — like the scratch bank, `last_flag_reasons`, `last_lq_refit_count`, `last_lq_tie_reasons` — per thread: read
    from another thread than the one that ran the fit they are 0, not an error."""
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().pmi_mle_last_refit_count(ctypes.byref(n), stream), "pmi_mle_last_refit_count")
    return int(n.value)


_prewarm_threads = {}      # (device, Y, X) -> the thread making that size's FFT plans


def prewarm_fft(Y: int, X: int) -> None:
    """Start making the FFT plans of RCC undrift for Y x X frames on a side thread (once per device and size): rocFFT
    compiles a plan's kernels when the plan is made — 2.5 s at 2048 x 2048 — and a caller that localizes first has that
    time.  Call it only when an undrift will follow (`localize_file(drift=...)` does).  The worker sets the CALLER's
    device before it makes the plans (plans belong to a device, and a new thread starts on device 0), is not a daemon,
    and is joined by `join_fft_prewarm` — before the correlations run, and at interpreter exit — so it never outlives
    the library it is calling into."""
    import threading
    L = _lib.load()
    dev = ctypes.c_int(0)
    _lib.check(L.pmi_get_device(ctypes.byref(dev)), "pmi_get_device")
    key = (int(dev.value), int(Y), int(X))
    if key in _prewarm_threads or min(key[1:]) < 64:
        return

    def work():
        try:
            if L.pmi_set_device(key[0]) == 0:
                L.pmi_fft_prewarm(key[1], key[2])       # ctypes releases the GIL
        except Exception:      # noqa: BLE001 - a convenience: the correlation makes its plans itself if this did not
            pass
    t = threading.Thread(target=work, name="pmi-fft-prewarm", daemon=False)
    _prewarm_threads[key] = t
    t.start()


def join_fft_prewarm() -> None:
    """Wait for the plan-making threads (a finished thread stays in the table: its size is not made again)."""
    for t in list(_prewarm_threads.values()):
        if t.is_alive():
            t.join()


import atexit as _atexit      # noqa: E402
_atexit.register(join_fft_prewarm)


def last_lq_refit_count() -> int:
    """Spots the last least-squares call fitted a second time with MINPACK's summation order."""
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().pmi_gausslq_last_refit_count(ctypes.byref(n)), "pmi_gausslq_last_refit_count")
    return int(n.value)


LQ_MODES = {"fast": 0, "refit": 1, "strict": 2}


def set_lq_mode(mode: str = "strict"):
    """How the sums over the residual rows of the least-squares fit run (pmi_gausslq_set_mode): "fast" = tree sums
    only; "refit" = tree sums, spots with a decision of lmdif near its threshold fitted again in MINPACK's order;
    "strict" (the library's default, and this function's) = every spot in MINPACK's order (scipy.optimize.leastsq at
    picasso/gausslq.py:240-242: the oracle's bits)."""
    if mode not in LQ_MODES:
        raise ValueError(f"unknown gausslq mode {mode!r}")
    _lib.check(_lib.load().pmi_gausslq_set_mode(LQ_MODES[mode]), "pmi_gausslq_set_mode")


def get_lq_mode() -> str:
    m = ctypes.c_int(0)
    _lib.check(_lib.load().pmi_gausslq_get_mode(ctypes.byref(m)), "pmi_gausslq_get_mode")
    return {v: k for k, v in LQ_MODES.items()}[m.value]


LQ_TIE_REASONS = ("pivot", "lmpar", "fnorm", "ratio", "ftol", "noise", "xtol", "fragile", "rounds_first_pass", "rounds_second_pass")


def last_lq_tie_reasons() -> dict:
    c = (ctypes.c_int64 * len(LQ_TIE_REASONS))()
    _lib.check(_lib.load().pmi_gausslq_last_tie_reasons(c, len(LQ_TIE_REASONS)), "pmi_gausslq_last_tie_reasons")
    return {k: int(v) for k, v in zip(LQ_TIE_REASONS, c)}


FLAG_REASONS = ("margin", "curvature", "narrow", "swing", "wild", "slow", "unstable")


def last_flag_reasons(stream=None) -> dict:
    """Of the spots the last MLE call fitted a second time, how many each criterion flagged (a spot can carry several)."""
    c = (ctypes.c_int64 * len(FLAG_REASONS))()
    _lib.check(_lib.load().pmi_mle_last_flag_reasons(c, len(FLAG_REASONS), stream), "pmi_mle_last_flag_reasons")
    return {k: int(v) for k, v in zip(FLAG_REASONS, c)}


def gaussmle_arrays(spots: np.ndarray, eps: float, max_it: int, method: str = "sigmaxy"):
    """The allocation contract of picasso/gaussmle.py:455-459."""
    if method not in _lib.MLE_METHODS:
        raise ValueError("Method not available.")
    _lib.require_gpu()
    spots = np.ascontiguousarray(spots, np.float32)
    if spots.ndim != 3 or spots.shape[1] != spots.shape[2]:
        raise ValueError("spots must have shape (N, box, box)")
    N, box, _ = spots.shape
    thetas = np.zeros((N, 6), np.float32)
    crlbs = np.full((N, 6), np.inf, np.float32)
    loglik = np.zeros(N, np.float32)
    iterations = np.zeros(N, np.int32)
    with _lib.lock():
        rc = _lib.load().pmi_gaussmle(_lib.ptr(spots), N, int(box), float(eps), int(max_it),
                                      _lib.MLE_METHODS[method], _lib.ptr(thetas), _lib.ptr(crlbs),
                                      _lib.ptr(loglik), _lib.ptr(iterations))
    _lib.check(rc, "pmi_gaussmle")
    return thetas, crlbs, loglik, iterations


def gausslq_arrays(spots: np.ndarray, full_output: bool = False):
    """theta (N,6) float32 as picasso/gausslq.py:247-268 fit_spots returns it; with
    full_output also MINPACK's info code and nfev per spot."""
    _lib.require_gpu()
    spots = np.ascontiguousarray(spots, np.float32)
    if spots.ndim != 3 or spots.shape[1] != spots.shape[2]:
        raise ValueError("spots must have shape (N, box, box)")
    N, box, _ = spots.shape
    theta = np.empty((N, 6), np.float32)
    info = np.zeros(N, np.int32)
    nfev = np.zeros(N, np.int32)
    with _lib.lock():
        rc = _lib.load().pmi_gausslq(_lib.ptr(spots), N, int(box), _lib.ptr(theta), _lib.ptr(info), _lib.ptr(nfev))
    _lib.check(rc, "pmi_gausslq")
    if full_output:
        return theta, info, nfev
    return theta


def avgroi_array(spots: np.ndarray) -> np.ndarray:
    _lib.require_gpu()
    spots = np.ascontiguousarray(spots, np.float32)
    N, box, _ = spots.shape
    theta = np.empty((N, 6), np.float32)
    theta.fill(np.nan)
    with _lib.lock():
        rc = _lib.load().pmi_avgroi(_lib.ptr(spots), N, int(box), _lib.ptr(theta))
    _lib.check(rc, "pmi_avgroi")
    return theta


def zfit_arrays(sx, sy, cx, cy):
    """-> z (before magnification) and squared calibration residual, float64."""
    _lib.require_gpu()
    sx = np.ascontiguousarray(sx, np.float32)
    sy = np.ascontiguousarray(sy, np.float32)
    cx = np.ascontiguousarray(cx, np.float64)
    cy = np.ascontiguousarray(cy, np.float64)
    if cx.shape != (7,) or cy.shape != (7,):
        raise ValueError("calibration needs 7 coefficients per axis")
    N = len(sx)
    z = np.zeros(N, np.float64)
    sq = np.zeros(N, np.float64)
    with _lib.lock():
        rc = _lib.load().pmi_zfit(_lib.ptr(sx), _lib.ptr(sy), N, _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(z), _lib.ptr(sq))
    _lib.check(rc, "pmi_zfit")
    return z, sq


def render_arrays(x, y, oversampling, y_min, x_min, y_max, x_max, lpx=None, lpy=None, min_blur_width=0.0, iso=False):
    """-> (n, image float32).  lpx/lpy None = histogram, else the Gaussian render."""
    _lib.require_gpu()
    L = _lib.load()
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    N = len(x)
    ny, nx = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(L.pmi_render_dims(float(oversampling), float(y_min), float(x_min), float(y_max), float(x_max),
                                 ctypes.byref(ny), ctypes.byref(nx)), "pmi_render_dims")
    if ny.value <= 0 or nx.value <= 0:
        raise ValueError("empty viewport")
    image = np.empty((ny.value, nx.value), np.float32)
    n = ctypes.c_int64(0)
    with _lib.lock():
        if lpx is None:
            rc = L.pmi_render_hist(_lib.ptr(x), _lib.ptr(y), N, float(oversampling), float(y_min), float(x_min),
                                   float(y_max), float(x_max), _lib.ptr(image), ny.value, nx.value, ctypes.byref(n))
        else:
            lpx = np.ascontiguousarray(lpx, np.float32)
            lpy = np.ascontiguousarray(lpy, np.float32)
            rc = L.pmi_render_gaussian(_lib.ptr(x), _lib.ptr(y), _lib.ptr(lpx), _lib.ptr(lpy), N, float(oversampling),
                                       float(y_min), float(x_min), float(y_max), float(x_max), float(min_blur_width),
                                       int(bool(iso)), _lib.ptr(image), ny.value, nx.value, ctypes.byref(n))
    _lib.check(rc, "pmi_render")
    return int(n.value), image


def xcorr_array(image_a, image_b) -> np.ndarray:
    _lib.require_gpu()
    a = np.ascontiguousarray(image_a, np.float64)
    b = np.ascontiguousarray(image_b, np.float64)
    if a.ndim != 2 or a.shape != b.shape:
        raise ValueError("images must be 2-D and of the same shape")
    out = np.empty_like(a)
    with _lib.lock():
        rc = _lib.load().pmi_xcorr(_lib.ptr(a), _lib.ptr(b), a.shape[0], a.shape[1], _lib.ptr(out))
    _lib.check(rc, "pmi_xcorr")
    return out


def rcc_pairs_arrays(segments, roi, box: int, pairs=None):
    """Pairs (i, j) of the segment images (all i < j when `pairs` is None) -> peak (n_pairs, 2),
    valid (n_pairs), fit windows (n_pairs, box, box) float64 and the crop offsets (Y_, X_)."""
    _lib.require_gpu()
    seg = np.ascontiguousarray(segments, np.float64)
    if seg.ndim != 3:
        raise ValueError("segments must have shape (n, Y, X)")
    n, Y, X = seg.shape
    if pairs is None:
        pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    pairs = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    n_pairs = len(pairs)
    peak = np.zeros((n_pairs, 2), np.int32)
    valid = np.zeros(n_pairs, np.int32)
    rois = np.zeros((n_pairs, box, box), np.float64)
    crop = np.zeros(2, np.int32)
    with _lib.lock():
        rc = _lib.load().pmi_rcc_pair_list(_lib.ptr(seg), n, Y, X, int(roi) if roi is not None else 0, int(box),
                                           _lib.ptr(pairs), n_pairs, _lib.ptr(peak), _lib.ptr(valid), _lib.ptr(rois),
                                           _lib.ptr(crop))
    _lib.check(rc, "pmi_rcc_pair_list")
    return peak, valid, rois, (int(crop[0]), int(crop[1]))


def peak_fit_arrays(rois):
    """(n, box, box) float64 correlation windows -> popt (n, 5) = a, xc, yc, s, b and the termination status of the
    bounded Gaussian fit (pmi_peak_fit; scipy's curve_fit in the reference, picasso/imageprocess.py:121-141)."""
    _lib.require_gpu()
    rois = np.ascontiguousarray(rois, np.float64)
    if rois.ndim != 3 or rois.shape[1] != rois.shape[2]:
        raise ValueError("rois must have shape (n, box, box)")
    n, box, _ = rois.shape
    popt = np.zeros((n, 5), np.float64)
    status = np.zeros(n, np.int32)
    with _lib.lock():
        rc = _lib.load().pmi_peak_fit(_lib.ptr(rois), n, int(box), _lib.ptr(popt), _lib.ptr(status))
    _lib.check(rc, "pmi_peak_fit")
    return popt, status


def rcc_shifts_arrays(segments, roi, box: int, pairs=None):
    """get_image_shift (picasso/imageprocess.py:53-161) for pairs (i, j) of the segment images (all i < j when
    `pairs` is None), entirely on the device -> shifts (n_pairs, 2) = (-yc, -xc), fit status (n_pairs)."""
    _lib.require_gpu()
    seg = np.ascontiguousarray(segments, np.float64)
    if seg.ndim != 3:
        raise ValueError("segments must have shape (n, Y, X)")
    n, Y, X = seg.shape
    if pairs is None:
        pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    pairs = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    n_pairs = len(pairs)
    shifts = np.zeros((n_pairs, 2), np.float64)
    status = np.zeros(n_pairs, np.int32)
    with _lib.lock():
        rc = _lib.load().pmi_rcc_shifts(_lib.ptr(seg), n, Y, X, int(roi) if roi is not None else 0, int(box), _lib.ptr(pairs),
                                        n_pairs, _lib.ptr(shifts), _lib.ptr(status))
    _lib.check(rc, "pmi_rcc_shifts")
    return shifts, status


class DeviceMovie:
    """A movie resident in HBM (pmi_malloc), for repeated calls without H2D."""

    def __init__(self, movie: np.ndarray):
        _lib.require_gpu()
        movie = as_movie_array(movie)
        self.capacity = max(movie.nbytes, 1)
        self._ptr = ctypes.c_void_p()
        _lib.check(_lib.load().pmi_malloc(ctypes.byref(self._ptr), self.capacity), "pmi_malloc")
        self.load(movie)

    def load(self, movie: np.ndarray):
        """Upload another stack of frames into the same allocation (it must fit)."""
        movie = as_movie_array(movie)
        if movie.nbytes > self.capacity:
            raise ValueError("DeviceMovie.load: stack larger than the allocation")
        self.shape = movie.shape
        self.dtype = movie.dtype
        self.nbytes = movie.nbytes
        _lib.check(_lib.load().pmi_memcpy_h2d(self._ptr, _lib.ptr(movie), self.nbytes), "pmi_memcpy_h2d")

    @property
    def ptr(self):
        return self._ptr

    def free(self):
        if self._ptr:
            _lib.load().pmi_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceWorkspace:
    """Grow-only device buffers of the fused pipelines (table + row count), kept across submissions so that a
    chunked run does not allocate and free per chunk (hipFree waits for the whole device, uploads included)."""

    def __init__(self):
        self._table = ctypes.c_void_p()
        self._bytes = 0
        self._dn = ctypes.c_void_p()

    def table(self, nbytes: int):
        L = _lib.load()
        if nbytes > self._bytes:
            if self._table:
                L.pmi_free(self._table)
                self._table, self._bytes = ctypes.c_void_p(), 0
            want = nbytes + nbytes // 4
            _lib.check(L.pmi_malloc(ctypes.byref(self._table), want), "pmi_malloc")
            self._bytes = want
        return self._table

    def count(self):
        if not self._dn:
            _lib.check(_lib.load().pmi_malloc(ctypes.byref(self._dn), 8), "pmi_malloc")
        return self._dn

    def free(self):
        L = _lib.load()
        if self._table:
            L.pmi_free(self._table)
        if self._dn:
            L.pmi_free(self._dn)
        self._table, self._bytes, self._dn = ctypes.c_void_p(), 0, ctypes.c_void_p()


class DeviceStream:
    """A non-blocking HIP stream of the library (pmi_stream_create)."""

    def __init__(self):
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().pmi_stream_create(ctypes.byref(self.handle)), "pmi_stream_create")

    def destroy(self):
        if self.handle:
            _lib.load().pmi_stream_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def _localize_device(call, columns, d_movie_ptr, dtype, shape, roi, frame_bounds, cap, stream, f_lo, f_hi, work=None):
    """Shared driver of the fused pipelines: allocate the table (or take it from `work`), submit, grow on
    overflow, copy the columns back (on `stream` when one is given, so that nothing orders against the
    default stream)."""
    _lib.require_gpu()
    L = _lib.load()
    F, Y, X = shape
    r = normalise_roi(roi, Y, X)
    lo, hi = frame_range(frame_bounds, F)
    if f_lo is not None:
        lo = max(lo, f_lo)
    if f_hi is not None:
        hi = min(hi, f_hi)
    cap = int(cap or max(4096, 256 * F))
    ncol = len(columns)
    own = work is None
    ws = DeviceWorkspace() if own else work

    def fetch(dst, src, nbytes):
        if stream is None:
            _lib.check(L.pmi_memcpy_d2h(_lib.ptr(dst), src, nbytes), "d2h")
        else:
            _lib.check(L.pmi_memcpy_d2h_async(_lib.ptr(dst), src, nbytes, stream), "d2h")

    try:
        while True:
            table = ws.table(ncol * cap * 4)
            dn = ws.count()
            n = np.zeros(1, np.int64)
            with _lib.lock():
                call(L, d_movie_ptr, dtype_code(dtype), F, Y, X, r, lo, hi, table, cap, dn, stream)
                fetch(n, dn, 8)
                _lib.check(L.pmi_stream_synchronize(stream), "sync")
            n = int(n[0])
            if n > cap:
                cap = n
                continue
            out = {}
            for c, (name, dt) in enumerate(columns):
                out[name] = np.empty(n, dt)
                if n:
                    fetch(out[name], ctypes.c_void_p(table.value + c * cap * 4), n * 4)
            if stream is not None and n:
                _lib.check(L.pmi_stream_synchronize(stream), "sync")
            return out
    finally:
        if own:
            ws.free()


def localize_mle_device(d_movie_ptr, dtype, shape, box, min_ng, camera, eps=1e-3, max_it=100,
                        method="sigmaxy", roi=None, frame_bounds=None, cap=None, stream=None,
                        f_lo=None, f_hi=None, work=None):
    """identify -> fused cut+fit -> table on a resident movie.  Returns a dict of
    numpy columns (LOC_COLUMNS).  d_movie_ptr: int / c_void_p device address."""
    def call(L, d_movie, code, F, Y, X, r, lo, hi, table, cap_, dn, stream_):
        rc = L.pmi_localize_mle_dev(d_movie, code, F, Y, X, int(box), float(min_ng), _lib.ptr(r), lo, hi,
                                    float(camera["Baseline"]), float(camera["Sensitivity"]), float(camera["Gain"]),
                                    float(eps), int(max_it), _lib.MLE_METHODS[method], table, cap_, dn, stream_)
        _lib.check(rc, "pmi_localize_mle_dev")
    return _localize_device(call, LOC_COLUMNS, d_movie_ptr, dtype, shape, roi, frame_bounds, cap, stream, f_lo, f_hi, work)


def localize_lq_device(d_movie_ptr, dtype, shape, box, min_ng, camera, roi=None, frame_bounds=None, cap=None,
                       stream=None, f_lo=None, f_hi=None, work=None):
    """identify -> fused cut + least-squares fit -> 11-column table (LQ_COLUMNS)."""
    em = int(camera["Gain"] > 1)

    def call(L, d_movie, code, F, Y, X, r, lo, hi, table, cap_, dn, stream_):
        rc = L.pmi_localize_lq_dev(d_movie, code, F, Y, X, int(box), float(min_ng), _lib.ptr(r), lo, hi,
                                   float(camera["Baseline"]), float(camera["Sensitivity"]), float(camera["Gain"]),
                                   em, table, cap_, dn, stream_)
        _lib.check(rc, "pmi_localize_lq_dev")
    return _localize_device(call, LQ_COLUMNS, d_movie_ptr, dtype, shape, roi, frame_bounds, cap, stream, f_lo, f_hi, work)
