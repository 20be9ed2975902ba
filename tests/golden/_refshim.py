"""Shim loader for the reference's own source files (golden minting only).

TEST INFRASTRUCTURE. Runs ONLY in the build container, where the reference
tree is mounted read-only at /root/reference. Nothing here travels to the GPU
box as reference code: this module loads the reference's ``gaussmle.py``,
``gausslq.py``, ``avgroi.py``, ``zfit.py``, ``localize.py``, ``render.py`` and
``imageprocess.py`` *from where they lie* and executes them as plain Python behind stand-in modules for the
packages this image lacks (numba, dask, h5py/Qt-dependent ``picasso.lib`` /
``picasso.io``).  See SURVEY.md section 8c for why the real numba path cannot
run here.

Consequence (stated in DESIGN.md): goldens minted through this shim follow
NumPy-2 (NEP 50) scalar promotion, not numba's.  Converged parameters agree
with numba's float64-intermediate arithmetic to ~1e-5 px; iteration counts may
differ by one on borderline convergence.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = os.environ.get("PICASSO_REFERENCE", "/root/reference")


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn

    return wrap


def _vectorize(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return np.vectorize(args[0])

    def wrap(fn):
        return np.vectorize(fn)

    return wrap


def load_reference():
    """Return dict of reference modules executed from /root/reference."""
    if not os.path.isdir(os.path.join(REF, "picasso")):
        raise RuntimeError(f"reference tree not found at {REF}")
    if "picasso.localize" in sys.modules and getattr(
        sys.modules["picasso"], "_is_shim", False
    ):
        m = sys.modules
        return {
            k: m["picasso." + k]
            for k in ("gaussmle", "gausslq", "avgroi", "zfit", "localize", "render", "imageprocess")
        }

    numba = types.ModuleType("numba")
    numba.jit = _identity_decorator
    numba.njit = _identity_decorator
    numba.vectorize = _vectorize
    numba.prange = range
    sys.modules["numba"] = numba

    dask = types.ModuleType("dask")
    dask_array = types.ModuleType("dask.array")
    dask.array = dask_array
    sys.modules["dask"] = dask
    sys.modules["dask.array"] = dask_array

    pkg = types.ModuleType("picasso")
    pkg.__path__ = [os.path.join(REF, "picasso")]
    pkg.__version__ = "0.10.3"
    pkg._is_shim = True
    sys.modules["picasso"] = pkg

    lib = types.ModuleType("picasso.lib")

    def deprecation_warning(message):
        warnings.warn(message, DeprecationWarning, stacklevel=2)

    def n_futures_done(futures):
        return sum(f.done() for f in futures)

    def get_from_metadata(info, key, default=None, *, raise_error=False):
        if isinstance(info, dict):
            info = [info]
        for d in reversed(info):
            if key in d:
                return d[key]
        if raise_error:
            raise KeyError(key)
        return default

    def ensure_sanity(locs, info):
        # behaviour of picasso/lib.py:1786-1832 restated for the zfit goldens
        locs = locs.copy()
        locs.replace([np.inf, -np.inf], np.nan, inplace=True)
        locs.dropna(axis=0, how="any", inplace=True)
        width = get_from_metadata(info, "Width")
        height = get_from_metadata(info, "Height")
        locs = locs[locs.x < width]
        locs = locs[locs.y < height]
        for attr in ("x", "y", "lpx", "lpy", "lpz", "photons",
                     "ellipticity", "sx", "sy"):
            if attr in locs.columns:
                locs = locs[locs[attr] >= 0]
        return locs

    lib.deprecation_warning = deprecation_warning
    lib.n_futures_done = n_futures_done
    lib.get_from_metadata = get_from_metadata
    lib.ensure_sanity = ensure_sanity
    lib.__getattr__ = lambda name: object  # annotation-only names
    sys.modules["picasso.lib"] = lib
    pkg.lib = lib

    io = types.ModuleType("picasso.io")

    class AbstractPicassoMovie:  # isinstance target only
        def __init__(self):
            pass

    class ND2Movie(AbstractPicassoMovie):
        use_dask = False

    io.AbstractPicassoMovie = AbstractPicassoMovie
    io.ND2Movie = ND2Movie
    io.load_user_settings = lambda: {"Localize": {"cpu_utilization": 0.8}}
    io.save_user_settings = lambda settings: None
    sys.modules["picasso.io"] = io
    pkg.io = io

    ext = types.ModuleType("picasso.ext")
    ext.__path__ = []
    bitplane = types.ModuleType("picasso.ext.bitplane")
    bitplane.IMSWRITER = False
    ext.bitplane = bitplane
    sys.modules["picasso.ext"] = ext
    sys.modules["picasso.ext.bitplane"] = bitplane
    pkg.ext = ext

    postprocess = types.ModuleType("picasso.postprocess")
    sys.modules["picasso.postprocess"] = postprocess
    pkg.postprocess = postprocess

    # stand-ins for GUI / image-file packages render.py and imageprocess.py import at module level
    class _StubMeta(type):
        def __getattr__(cls, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Stub

    class _Stub(metaclass=_StubMeta):        # callable, subclassable, any attribute
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Stub()

        def __call__(self, *a, **k):
            return _Stub()

    class _Anything(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            full = self.__name__ + "." + name
            if full in sys.modules:
                return sys.modules[full]
            return _Stub
    for modname in ("imageio", "imageio.v2", "PyQt6", "PyQt6.QtGui", "PyQt6.QtCore", "PyQt6.QtSvg", "PyQt6.QtWidgets"):
        if modname not in sys.modules:
            try:
                importlib.import_module(modname)
            except ImportError:
                sys.modules[modname] = _Anything(modname)
    import matplotlib
    matplotlib.use("Agg")

    out = {}
    for name in ("gaussmle", "gausslq", "avgroi", "zfit", "localize", "render", "imageprocess"):
        path = os.path.join(REF, "picasso", name + ".py")
        spec = importlib.util.spec_from_file_location("picasso." + name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["picasso." + name] = mod
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        out[name] = mod
    return out


def load_test_movie():
    """The reference's bundled 100x32x32 <u2 movie (tests/data/testdata.raw)."""
    path = os.path.join(REF, "tests", "data", "testdata.raw")
    return np.fromfile(path, dtype="<u2").reshape(100, 32, 32)
