"""ctypes binding of libpicasso_hip.so (the C ABI in include/picasso_hip.h).

The library is loaded lazily from the package directory, like the reference
loads Gpufit (picasso/ext/pygpufit/gpufit.py:24-37).  Unlike the reference there
is NO fallback: if the HIP library is missing or no GPU is visible, every
compute call raises.  The product path never routes through a CPU
implementation.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PICASSO_AMD_LIB") or os.path.join(_HERE, "libpicasso_hip.so")      # override: A/B builds

PMI_OK = 0
PMI_ERR_CAPACITY = 1
PMI_LOC_COLUMNS = 17
PMI_LQ_COLUMNS = 11
PMI_MAX_BOX = 21

DTYPE_CODES = {
    np.dtype("uint16"): 0, np.dtype("uint8"): 1, np.dtype("int16"): 2,
    np.dtype("uint32"): 3, np.dtype("int32"): 4, np.dtype("float32"): 5,
}
MLE_METHODS = {"sigma": 0, "sigmaxy": 1}

# every symbol include/picasso_hip.h declares: (name, restype, argtypes)
_i32, _i64, _f64, _p, _sz = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
SYMBOLS = {
    "pmi_version": (_i32, []),
    "pmi_last_error": (ctypes.c_char_p, []),
    "pmi_device_count": (_i32, []),
    "pmi_set_device": (_i32, [_i32]),
    "pmi_get_device": (_i32, [_p]),
    "pmi_device_info": (_i32, [_p, _sz, _p, _p]),
    "pmi_malloc": (_i32, [_p, _sz]),
    "pmi_free": (_i32, [_p]),
    "pmi_memcpy_h2d": (_i32, [_p, _p, _sz]),
    "pmi_memcpy_d2h": (_i32, [_p, _p, _sz]),
    "pmi_stream_synchronize": (_i32, [_p]),
    "pmi_stream_create": (_i32, [_p]),
    "pmi_stream_destroy": (_i32, [_p]),
    "pmi_memcpy_d2h_async": (_i32, [_p, _p, _sz, _p]),
    "pmi_release_scratch": (_i32, []),
    "pmi_scratch_bank": (_i32, [_i32]),
    "pmi_identify_set_narrow_chunk": (_i32, [_i64]),
    "pmi_identify": (_i32, [_p, _i32, _i64, _i64, _i64, _i32, _f64, _p, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "pmi_identify_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _i32, _f64, _p, _i64, _i64, _p, _p, _p, _p, _i64, _p, _p]),
    "pmi_net_gradient": (_i32, [_p, _i64, _i64, _p, _p, _i64, _i32, _p, _p, _p]),
    "pmi_get_spots": (_i32, [_p, _i32, _i64, _i64, _i64, _p, _p, _p, _i64, _i32, _f64, _f64, _f64, _p]),
    "pmi_get_spots_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _p, _p, _p, _i64, _p, _i32, _f64, _f64, _f64, _p, _p]),
    "pmi_gaussmle": (_i32, [_p, _i64, _i32, _f64, _i32, _i32, _p, _p, _p, _p]),
    "pmi_gaussmle_dev": (_i32, [_p, _i64, _p, _i32, _f64, _i32, _i32, _p, _p, _p, _p, _p]),
    "pmi_gaussmle_movie_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _p, _p, _p, _i64, _p, _i32, _f64, _f64, _f64,
                                      _f64, _i32, _i32, _p, _p, _p, _p, _p]),
    "pmi_mle_set_mode": (_i32, [_i32, _f64]),
    "pmi_mle_get_mode": (_i32, [_p, _p]),
    "pmi_mle_set_libm": (_i32, [_i32]),
    "pmi_mle_get_libm": (_i32, [_p]),
    "pmi_libm_eval_dev": (_i32, [_i32, _p, _i64, _p, _p]),
    "pmi_mle_last_refit_count": (_i32, [_p, _p]),
    "pmi_mle_last_flag_reasons": (_i32, [_p, _i32, _p]),
    "pmi_locs_from_fits_dev": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i32, _p, _p]),
    "pmi_localize_set_handoff": (_i32, [_i32]),
    "pmi_localize_set_ranges": (_i32, [_i32]),
    "pmi_localize_set_defer": (_i32, [_i32]),
    "pmi_localize_mle_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _i32, _f64, _p, _i64, _i64, _f64, _f64, _f64,
                                    _f64, _i32, _i32, _p, _i64, _p, _p]),
    "pmi_gausslq_set_mode": (_i32, [_i32]),
    "pmi_gausslq_get_mode": (_i32, [_p]),
    "pmi_gausslq_last_refit_count": (_i32, [_p]),
    "pmi_gausslq_last_tie_reasons": (_i32, [_p, _i32]),
    "pmi_gausslq": (_i32, [_p, _i64, _i32, _p, _p, _p]),
    "pmi_gausslq_dev": (_i32, [_p, _i64, _p, _i32, _p, _p, _p, _p]),
    "pmi_gausslq_movie_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _p, _p, _p, _i64, _p, _i32, _f64, _f64, _f64,
                                     _p, _p, _p, _p]),
    "pmi_locs_from_fits_lq_dev": (_i32, [_p, _p, _p, _p, _p, _i64, _p, _i32, _p, _p]),
    "pmi_localize_lq_dev": (_i32, [_p, _i32, _i64, _i64, _i64, _i32, _f64, _p, _i64, _i64, _f64, _f64, _f64,
                                   _i32, _p, _i64, _p, _p]),
    "pmi_zfit": (_i32, [_p, _p, _i64, _p, _p, _p, _p]),
    "pmi_zfit_dev": (_i32, [_p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "pmi_avgroi": (_i32, [_p, _i64, _i32, _p]),
    "pmi_avgroi_dev": (_i32, [_p, _i64, _p, _i32, _p, _p]),
    "pmi_render_dims": (_i32, [_f64, _f64, _f64, _f64, _f64, _p, _p]),
    "pmi_render_hist": (_i32, [_p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _p, _i64, _i64, _p]),
    "pmi_render_hist_dev": (_i32, [_p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _p, _i64, _i64, _p, _p]),
    "pmi_render_gaussian": (_i32, [_p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _i32, _p, _i64, _i64, _p]),
    "pmi_render_gaussian_dev": (_i32, [_p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _i32, _p, _i64, _i64, _p, _p]),
    "pmi_xcorr": (_i32, [_p, _p, _i64, _i64, _p]),
    "pmi_rcc_pairs": (_i32, [_p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p]),
    "pmi_rcc_pair_list": (_i32, [_p, _i64, _i64, _i64, _i64, _i32, _p, _i64, _p, _p, _p, _p]),
    "pmi_fft_prewarm": (_i32, [_i64, _i64]),
    "pmi_peak_fit": (_i32, [_p, _i64, _i32, _p, _p]),
    "pmi_rcc_shifts": (_i32, [_p, _i64, _i64, _i64, _i64, _i32, _p, _i64, _p, _p]),
    "pmi_comm_available": (_i32, []),
    "pmi_comm_unique_id": (_i32, [_p]),
    "pmi_comm_init": (_i32, [_p, _i32, _i32, _p]),
    "pmi_comm_info": (_i32, [_p, _p, _p]),
    "pmi_comm_library_path": (_i32, [_p, _sz]),
    "pmi_comm_destroy": (_i32, [_p]),
    "pmi_allgather_locs": (_i32, [_p, _p, _i32, _i64, _p, _p, _p, _p]),
    "pmi_compact_gathered_dev": (_i32, [_p, _p, _i32, _i32, _i64, _p, _i64, _p, _p]),
    "pmi_event_create": (_i32, [_p]),
    "pmi_event_record": (_i32, [_p, _p]),
    "pmi_event_elapsed_ms": (_i32, [_p, _p, _p]),
    "pmi_event_destroy": (_i32, [_p]),
    "pmi_last_scan_kernel": (_i32, [_p, ctypes.c_size_t]),
    "pmi_set_kernel_timing": (_i32, [_i32]),
    "pmi_last_kernel_ms": (_i32, [_p, _p]),
}

_lib = None
_lock = threading.Lock()      # one context per process, calls serialised — for every thread that never bound itself
_tls = threading.local()       # .key = (device, bank) of a thread bound by bind_thread
_bound_locks = {}
_bound_locks_mu = threading.Lock()


class HipBackendError(RuntimeError):
    """Raised when the HIP library reports an error (status != 0)."""


def load():
    """Load libpicasso_hip.so and bind every declared symbol (no GPU needed)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C picasso_amd/csrc`.  There is no CPU fallback.")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  Whichever loads first wins, and torch cannot see the GPU through the
        # system copy, so when torch is installed it must be imported before our library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)     # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return load().pmi_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != PMI_OK:
        raise HipBackendError(f"{what or 'libpicasso_hip'}: status {rc}: {last_error()}")


def device_count() -> int:
    return int(load().pmi_device_count())


def require_gpu():
    if device_count() < 1:
        raise HipBackendError("no HIP device visible: picasso_amd needs an AMD GPU (gfx950); "
                              "there is no CPU fallback")


def ptr(a):
    """void* of a numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.c_void_p)


def bind_thread(device: int, bank: int = 0) -> None:
    """Make `device` the calling thread's device (pmi_set_device: HIP keeps the current device per thread) and `bank` its
    scratch bank (pmi_scratch_bank).  Everything the library keeps on a device — scratch, tables, plans, side streams — is
    keyed by the device current in the calling thread, so threads bound to different devices (or to the two banks of one
    device) run side by side: `lock()` then serialises only threads bound alike.  A thread that never binds stays on HIP's
    default device, bank 0, behind the one process-wide lock as before."""
    L = load()
    check(L.pmi_set_device(int(device)), "pmi_set_device")
    check(L.pmi_scratch_bank(int(bank)), "pmi_scratch_bank")
    _tls.key = (int(device), int(bank))


def bound_to():
    """(device, bank) the calling thread bound itself to, or None."""
    return getattr(_tls, "key", None)


def current_key():
    """(device, bank) the calling thread's library calls use: what it bound itself to, else HIP's current device of
    the thread and bank 0 (None when no device can be asked, e.g. on a box without a GPU)."""
    key = getattr(_tls, "key", None)
    if key is not None:
        return key
    dev = ctypes.c_int(0)
    try:
        if load().pmi_get_device(ctypes.byref(dev)) != PMI_OK:
            return None
    except (ImportError, OSError):
        return None
    return (int(dev.value), 0)


def lock():
    """One lock per (device, scratch bank): a thread that never bound itself takes the lock of the device it really
    runs on (bank 0), i.e. the same lock as a lane bound to that device and bank — the two write the same scratch."""
    key = current_key()
    if key is None:
        return _lock
    with _bound_locks_mu:
        lk = _bound_locks.get(key)
        if lk is None:
            lk = _bound_locks[key] = threading.Lock()
        return lk
