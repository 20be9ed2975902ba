"""Golden vectors for the remaining public functions of picasso/localize.py on the path, minted by executing the
reference's own source through the import shim (tests/golden/_refshim.py):

  picks_to_identifications, locs_to_identifications (host-side table constructors)
  locs_from_fits (the older 12-column MLE table)
  localize_3D on tests/data/testdata.raw with the conftest CALIB_3D calibration (NumPy-executed reference)

Run in the build container:  python tests/golden/make_goldens_surface.py
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

ref = _refshim.load_reference()
loc = ref["localize"]
rng = np.random.default_rng(20261002)
out = {}

# picks -> identifications, with and without drift
picks = [(12.5, 40.25), (100.0, 7.75), (63.1, 63.9)]
drift = pd.DataFrame({"x": rng.normal(0, 0.3, 17).cumsum(), "y": rng.normal(0, 0.3, 17).cumsum()})
for tag, kw in (("picks_plain", dict(n_frames=9)), ("picks_drift", dict(drift=drift))):
    df = loc.picks_to_identifications(picks, **kw)
    for c in df.columns:
        out[f"{tag}_{c}"] = df[c].to_numpy()
out["picks"] = np.asarray(picks)
out["drift_x"], out["drift_y"] = drift["x"].to_numpy(), drift["y"].to_numpy()

# locs -> identifications
locs = pd.DataFrame({"frame": rng.integers(0, 60, 25).astype(np.uint32), "x": rng.uniform(5, 50, 25).astype(np.float32),
                     "y": rng.uniform(5, 50, 25).astype(np.float32)})
df = loc.locs_to_identifications(locs, [{"Frames": 60}], 4)
for c in locs.columns:
    out[f"l2i_in_{c}"] = locs[c].to_numpy()
for c in df.columns:
    out[f"l2i_{c}"] = df[c].to_numpy()

# the older table builder
n = 40
ids = pd.DataFrame({"frame": np.sort(rng.integers(0, 12, n)), "x": rng.integers(4, 28, n), "y": rng.integers(4, 28, n),
                    "net_gradient": rng.uniform(5e3, 4e4, n).astype(np.float32)})
theta = rng.uniform(0.5, 6.5, (n, 6)).astype(np.float32)
crlb = rng.uniform(1e-4, 1e-2, (n, 6)).astype(np.float32)
ll = rng.normal(-200, 30, n).astype(np.float32)
it = rng.integers(3, 30, n).astype(np.int32)
df = loc.locs_from_fits(ids, theta, crlb, ll, it, 7)
for c in ids.columns:
    out[f"lff_ids_{c}"] = ids[c].to_numpy()
out["lff_theta"], out["lff_crlb"], out["lff_ll"], out["lff_it"] = theta, crlb, ll, it
out["lff_columns"] = np.array(list(df.columns))
for c in df.columns:
    out[f"lff_{c}"] = df[c].to_numpy()

# _local_maxima / _net_gradient / _gradient_at on single frames, including positions whose window touches
# row / column -1 (wraps to the last one) and a non-standard unit-vector table
tm = np.fromfile(os.path.join(_refshim.REF, "tests", "data", "testdata.raw"), dtype="<u2").reshape(100, 32, 32)
for tag, fidx, box in (("ng7", 12, 7), ("ng5", 40, 5), ("ng9", 77, 9)):
    frame = np.float32(tm[fidx])
    h = box // 2
    my, mx = loc._local_maxima(frame, box)
    yy = np.concatenate([my, [h, h, 20, 31 - h - 1]]).astype(np.int64)
    xx = np.concatenate([mx, [h, 18, h, 31 - h - 1]]).astype(np.int64)
    ux = np.zeros((box, box), np.float32); uy = np.zeros((box, box), np.float32)
    for i in range(box):
        val = h - i
        ux[:, i] = val
        uy[i, :] = val
    with np.errstate(invalid="ignore"):
        norm = np.sqrt(ux ** 2 + uy ** 2)
        ux /= norm
        uy /= norm
    out[f"{tag}_frame"], out[f"{tag}_y"], out[f"{tag}_x"] = frame, yy, xx
    out[f"{tag}_n_maxima"] = np.asarray(len(my))
    out[f"{tag}_ux"], out[f"{tag}_uy"] = ux, uy
    out[f"{tag}_ng"] = loc._net_gradient(frame, yy, xx, box, uy, ux)
    ruy = rng.normal(0, 1, (box, box)).astype(np.float32); rux = rng.normal(0, 1, (box, box)).astype(np.float32)
    out[f"{tag}_ruy"], out[f"{tag}_rux"] = ruy, rux
    out[f"{tag}_rng"] = loc._net_gradient(frame, yy, xx, box, ruy, rux)
    out[f"{tag}_grad"] = np.array([loc._gradient_at(frame, int(a), int(b), 0) for a, b in zip(yy, xx)], np.float32)

# localize_3D end to end (identify + gaussmle + zfit) on the reference's test movie
movie = np.memmap(os.path.join(_refshim.REF, "tests", "data", "testdata.raw"), dtype="<u2", mode="r", shape=(100, 32, 32))
info = [{"Frames": 100, "Height": 32, "Width": 32, "Data Type": "uint16", "Byte Order": "<"}]
zg = np.load(os.path.join(HERE, "zfit_calib3d.npz"))          # the conftest CALIB_3D coefficients
calib = {"X Coefficients": [float(v) for v in zg["cx"]], "Y Coefficients": [float(v) for v in zg["cy"]],
         "Magnification factor": 0.79}
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0, "Qe": 1.0, "Pixelsize": 130}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    l3, info3 = loc.localize_3D(movie, movie_info=info, camera_info=cam, box=7, minimum_ng=5000.0,
                                calibration_3d=dict(calib), fitting_method="gaussmle", multiprocess=False)
out["l3d_columns"] = np.array(list(l3.columns))
out["l3d_index"] = np.asarray(l3.index)
for c in l3.columns:
    out[f"l3d_{c}"] = l3[c].to_numpy()
out["l3d_cx"], out["l3d_cy"] = np.asarray(calib["X Coefficients"]), np.asarray(calib["Y Coefficients"])
out["l3d_mag"] = np.asarray(calib["Magnification factor"])
out["l3d_info_len"] = np.asarray(len(info3))
np.savez_compressed(os.path.join(HERE, "surface_cases.npz"), **out)
print("wrote surface_cases.npz:", len(out), "arrays;", len(l3), "3D localizations, columns", list(l3.columns))
