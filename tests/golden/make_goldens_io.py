#!/usr/bin/env python3
"""Fixtures for the wire-format row.  Needs h5py: run with /opt/conda/bin/python3.9.

  testdata_locs.hdf5 / .yaml   copies of the reference's own test DATA files (tests/data/), written by a
                               real Picasso run through h5py
  io_fixture.npz               the records of that file as real h5py reads them
  h5py_written_locs.hdf5       a table written here by h5py the way picasso/io.py:2104-2106 does
  io_fixture.npz: ours_read_by_h5py = 1 records that h5py could read back a file produced by
                               picasso_amd/_hdf5.py (structure, dtypes and values identical)
"""
import importlib.util
import os
import shutil
import sys
import tempfile

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PICASSO_REFERENCE", "/root/reference")
spec = importlib.util.spec_from_file_location("_hdf5", os.path.join(HERE, "..", "..", "picasso_amd", "_hdf5.py"))
_hdf5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(_hdf5)

for name in ("testdata_locs.hdf5", "testdata_locs.yaml"):
    shutil.copyfile(os.path.join(REF, "tests", "data", name), os.path.join(HERE, name))
with h5py.File(os.path.join(HERE, "testdata_locs.hdf5"), "r") as f:
    ref_locs = f["locs"][...]

rng = np.random.default_rng(3)
dt = np.dtype([("frame", "<u4"), ("x", "<f4"), ("y", "<f4"), ("photons", "<f4"), ("lpx", "<f4"), ("lpy", "<f4"),
               ("n_id", "<i8"), ("z", "<f8"), ("iterations", "<i4")])
rec = np.zeros(257, dt)
for n in dt.names:
    rec[n] = rng.integers(0, 300, len(rec)) if dt[n].kind in "iu" else rng.normal(5, 2, len(rec))
with h5py.File(os.path.join(HERE, "h5py_written_locs.hdf5"), "w") as f:
    f.create_dataset("locs", data=rec)

ok = 0
with tempfile.TemporaryDirectory() as tmp:
    p = os.path.join(tmp, "ours.hdf5")
    _hdf5.write(p, {"locs": rec, "identifications": rec[["frame", "x", "y"]][:7].copy(), "empty": rec[:0]})
    with h5py.File(p, "r") as f:
        ok = int(sorted(f.keys()) == ["empty", "identifications", "locs"] and f["locs"].dtype == rec.dtype
                 and np.array_equal(f["locs"][...], rec) and f["empty"].shape == (0,)
                 and np.array_equal(f["identifications"][...]["x"], rec["x"][:7]))
np.savez_compressed(os.path.join(HERE, "io_fixture.npz"), ref_locs=ref_locs, h5py_written=rec, ours_read_by_h5py=np.int64(ok))
print("reference records", ref_locs.shape, ref_locs.dtype.names, "| h5py reads our writer:", bool(ok))
