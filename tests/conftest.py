"""pytest configuration: markers, paths, shared golden loaders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def testdata_movie():
    """The reference's bundled 100x32x32 uint16 test movie (data fixture)."""
    return golden("testdata_movie")["movie"]


def roi_from(arr):
    return ((int(arr[0]), int(arr[1])), (int(arr[2]), int(arr[3]))) if len(arr) else None


def bounds_from(arr):
    return (int(arr[0]), int(arr[1])) if len(arr) else None


MLE_DATASETS = ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7",
                "degenerate7", "poisson9", "poisson13", "poisson5"]
# degenerate7 rows whose trajectory is chaotic / underflow-driven (corner hot pixel,
# pure noise): outputs depend on float32-vs-float64 underflow of exp(), so they are
# compared on iterations and finiteness only.  See DESIGN.md "degenerate spots".
DEGENERATE_LOOSE = {3, 4}


def assert_mle_rows(x, y, sx, sy, photons, it, ox, oy, osx, osy, ophotons, oit, max_it=100, label="", tol_px=1e-3):
    """North-star tolerance on EVERY row: the iteration count equals the reference's on every row, and wherever the
    reference converged (it < max_it) x, y, sigma agree to 1e-3 px and photons to 1e-2 relative.  No row is masked
    out: a spot whose convergence test is decided within rounding distance of eps is re-fitted on the device in the
    reference's arithmetic (csrc/gaussmle_strict.hip), so the counts must be equal, not merely close.  tol_px: 1e-3
    (the north star's) unless the caller runs a coarser convergence criterion eps, which leaves the converged
    position undetermined to eps."""
    it = np.asarray(it).astype(np.int64); oit = np.asarray(oit).astype(np.int64)
    differ = np.flatnonzero(it != oit)
    assert len(differ) == 0, f"{label}: iteration count differs on {len(differ)} of {len(it)} rows, first {differ[:5]}"
    conv = oit < max_it
    if conv.any():
        for name, a, b in (("x", x, ox), ("y", y, oy), ("sx", sx, osx), ("sy", sy, osy)):
            d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))[conv]
            assert not np.isnan(d).any() or np.array_equal(np.isnan(np.asarray(a)[conv]), np.isnan(np.asarray(b)[conv])), name
            assert np.nanmax(d, initial=0.0) < tol_px, f"{label}: {name} differs by {np.nanmax(d)} px on row {int(np.flatnonzero(conv)[np.nanargmax(d)])}"
        rel = (np.abs(np.asarray(photons, np.float64) - ophotons) / np.maximum(np.abs(ophotons), 1.0))[conv]
        assert np.nanmax(rel, initial=0.0) < 1e-2, f"{label}: photons differ by {np.nanmax(rel)} relative"
