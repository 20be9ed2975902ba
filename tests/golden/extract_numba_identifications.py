#!/usr/bin/env python3
"""Extract the real-numba identification pins from the reference's bundled table.

TEST INFRASTRUCTURE, build container only.  Needs h5py, which here exists only
in the Anaconda interpreter:

    /opt/conda/bin/python3.9 tests/golden/extract_numba_identifications.py

``tests/data/testdata_locs.hdf5`` in the reference was written by a real
(numba) Picasso run over a 1000-frame movie whose first 100 frames are the
bundled ``testdata.raw`` (box 7, min. net gradient 5000).  Its ``frame`` and
``net_gradient`` columns on frames 0-99 are therefore outputs of the numba
identify path and pin I1-I4 bit-for-bit.  The fit columns come from an older
fitter and are NOT goldens (kept only as loose sanity values).
"""
import os
import h5py
import numpy as np

REF = os.environ.get("PICASSO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
with h5py.File(os.path.join(REF, "tests", "data", "testdata_locs.hdf5"), "r") as f:
    locs = f["locs"][...]
sel = locs[locs["frame"] < 100]
np.savez_compressed(
    os.path.join(HERE, "numba_identifications_testdata.npz"),
    frame=sel["frame"].astype(np.int64),
    net_gradient=sel["net_gradient"].astype(np.float32),
    x_fit=sel["x"].astype(np.float32),
    y_fit=sel["y"].astype(np.float32),
)
print(len(sel), "rows")
