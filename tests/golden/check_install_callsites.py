"""Drop-in check at the reference's own call sites (build container only: needs /root/reference, no GPU).

The reference's modules are executed through the import shim (_refshim.py), ``picasso_amd.localize.install()``
rebinds their workers, and the reference's OWN ``localize.localize`` / ``fit2D`` / ``zfit.zfit`` /
``imageprocess.rcc`` are called.  Every device entry point of picasso_amd.backend is replaced by the CPU oracle
(no GPU here), so what this proves is the contract between the reference's call sites and the rebound functions:
argument lists, keyword names, return shapes, DataFrame columns/dtypes, metadata — and that the numbers the
reference then assembles equal the ones minted from the unmodified reference (tests/golden/*.npz).

    python tests/golden/check_install_callsites.py        -> exit 0, prints "call sites ok"
"""
import os
import sys
import types
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refshim  # noqa: E402

ref = _refshim.load_reference()
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend, localize as amd  # noqa: E402


# ---- the device entry points, answered by the oracle --------------------------------------------------
def identify_arrays(movie, min_ng, box, roi=None, frame_bounds=None, f_lo=None, f_hi=None):
    movie = backend.as_movie_array(movie)
    fr, y, x, ng = orc.identify(movie, min_ng, box, roi=roi, frame_bounds=frame_bounds, threads=4)
    keep = np.ones(len(fr), bool)
    if f_lo is not None:
        keep &= fr >= f_lo
    if f_hi is not None:
        keep &= fr <= f_hi
    return fr[keep].astype(np.int32), y[keep].astype(np.int32), x[keep].astype(np.int32), ng[keep]


def get_spots_array(movie, frame, y, x, box, baseline, sensitivity, gain):
    return orc.get_spots(backend.as_movie_array(movie), np.asarray(frame), np.asarray(y), np.asarray(x), box,
                         {"Baseline": baseline, "Sensitivity": sensitivity, "Gain": gain})


backend.identify_arrays = identify_arrays
backend.get_spots_array = get_spots_array
backend.gaussmle_arrays = lambda spots, eps, max_it, method="sigmaxy": orc.gaussmle(spots, eps, max_it, method, threads=4)
backend.gausslq_arrays = lambda spots, full_output=False: orc.gausslq(spots, threads=4)
backend.zfit_arrays = lambda sx, sy, cx, cy: orc.zfit(np.asarray(sx, np.float32), np.asarray(sy, np.float32), cx, cy, threads=4)
backend.avgroi_array = lambda spots: orc.avgroi(spots)


class _NoStream:                     # localize() must go through identify + fit2D here, not the fused device path
    def __init__(self, *a, **k):
        raise AssertionError("the reference's localize() does not know localize_streamed")


amd.install(ref["localize"], ref["gaussmle"], ref["gausslq"], ref["zfit"], ref["render"], ref["imageprocess"],
            types.SimpleNamespace())
L, Z = ref["localize"], ref["zfit"]
assert L.identify is amd.identify and L._fit2d_gaussmle is amd._fit2d_gaussmle and Z._fit_z.__module__ == "picasso_amd.zfit"

raw = os.path.join(_refshim.REF, "tests", "data", "testdata.raw")
movie = np.memmap(raw, dtype="<u2", mode="r", shape=(100, 32, 32))
info = [{"Frames": 100, "Height": 32, "Width": 32}]
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0, "Qe": 1.0, "Pixelsize": 130}
g = np.load(os.path.join(HERE, "surface_cases.npz"))
zg = np.load(os.path.join(HERE, "zfit_calib3d.npz"))
calib = {"X Coefficients": [float(v) for v in zg["cx"]], "Y Coefficients": [float(v) for v in zg["cy"]],
         "Magnification factor": 0.79}

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    # the reference's localize_3D -> its localize -> (rebound) identify, its fit2D -> (rebound) get_spots,
    # _fit2d_gaussmle; its zfit -> (rebound) _fit_z
    locs, out_info = L.localize_3D(movie, movie_info=info, camera_info=cam, box=7, minimum_ng=5000.0,
                                   calibration_3d=dict(calib), fitting_method="gaussmle", multiprocess=False)
assert list(locs.columns) == list(g["l3d_columns"]) and len(out_info) == int(g["l3d_info_len"]), list(locs.columns)
assert list(locs.index) == list(g["l3d_index"])
for c in locs.columns:
    a, b = locs[c].to_numpy(), g[f"l3d_{c}"]
    assert a.dtype == b.dtype, (c, a.dtype, b.dtype)
    if c in ("frame", "net_gradient"):
        assert np.array_equal(a, b), c
for c, tol in (("x", 1e-3), ("y", 1e-3), ("sx", 1e-3), ("sy", 1e-3), ("z", 0.5), ("d_zcalib", 2e-3)):
    assert np.max(np.abs(locs[c].to_numpy() - g[f"l3d_{c}"])) < tol, c
assert np.max(np.abs(locs["photons"].to_numpy() - g["l3d_photons"]) / g["l3d_photons"]) < 1e-2

# progress / abort contracts through the reference's fit2D and identify
seen = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ids, id_info = L.identify(movie, 5000.0, 7, threaded=True, progress_callback=seen.append, return_info=True)
    assert len(ids) == 30 and seen and id_info["Box Size"] == 7
    for method in ("gaussmle", "gausslq", "avg"):
        t, fit_info = L.fit2D(movie, info, dict(cam), ids, 7, fitting_method=method, multiprocess=(method != "avg"))
        assert len(t) == 30 and fit_info["Fit method"] == method, method
    none, _ = L.fit2D(movie, info, dict(cam), ids, 7, fitting_method="gaussmle", abort_callback=lambda: True)
    assert none is None
    assert L.identify(movie, 5000.0, 7, threaded=True, abort_callback=lambda: True, return_info=False) is None

# the older asynchronous entry points as the reference composes them
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    current, thetas, crlbs, lls, its = L.fit_async(movie, cam, ids, 7)
    amd.gaussmle.wait_async(current) if hasattr(amd.gaussmle, "wait_async") else None
    fs = ref["gausslq"].fit_spots_parallel(L.get_spots(movie, ids, 7, cam), asynch=True)
    assert ref["gausslq"].fits_from_futures(fs).shape == (30, 6)
print("call sites ok:", len(locs), "3D localizations through the reference's localize_3D with rebound workers")
