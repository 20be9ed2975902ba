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
