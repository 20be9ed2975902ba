"""CPU tier, build container only: the committed ``*_nbp.npz`` goldens regenerate from the reference tree.

A sample of every file's rows is minted again with ``tests/golden/_nbemu.py`` (the reference's own gaussmle.py / gausslq.py
executed with numba's typing rules) under THIS interpreter — NumPy 2.x, where nothing of the legacy scalar promotion the
files were minted under (/opt/conda's NumPy 1.26.4) helps: the emulator's rules alone must give the same bits.  Skipped
where the read-only reference tree does not exist (the GPU box)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/picasso"), reason="reference tree not present")


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, GOLDEN)
    import _nbemu
    return _nbemu


@pytest.mark.parametrize("name,rows", [("poisson7", (0, 17, 101)), ("degenerate7", (0, 3, 4, 5)), ("poisson13", (7,)), ("poisson5", (2, 30))])
def test_gaussmle_nbp_rows_regenerate(emu, name, rows):
    g = emu.load("gaussmle")
    d, ref = golden("gaussmle_" + name), golden("gaussmle_" + name + "_nbp")
    for method in ("sigmaxy", "sigma"):
        for i in rows:
            emu.ZERO_DIVISIONS[0] = 0
            with np.errstate(all="ignore"):
                th, cr, ll, it = g.gaussmle(d["spots"][i][None], 1e-3, 100, method)
            assert np.array_equal(th[0], ref[method + "_theta"][i], equal_nan=True), (name, method, i)
            assert it[0] == ref[method + "_iterations"][i]
            assert np.array_equal(ll[0], ref[method + "_loglik"][i], equal_nan=True)
            assert emu.ZERO_DIVISIONS[0] == ref[method + "_zero_division"][i]


def test_gaussmle_nbp_variants_regenerate(emu):
    g = emu.load("gaussmle")
    d, ref = golden("gaussmle_conftest_noisy"), golden("gaussmle_conftest_noisy_nbp")
    for tag, eps, max_it in (("_it3", 1e-3, 3), ("_eps5", 1e-5, 100)):
        th, cr, ll, it = g.gaussmle(d["spots"][5][None], eps, max_it, "sigmaxy")
        assert np.array_equal(th[0], ref["sigmaxy" + tag + "_theta"][5]) and it[0] == ref["sigmaxy" + tag + "_iterations"][5]


@pytest.mark.parametrize("name,rows", [("poisson7", (3, 150)), ("poisson13", (11,)), ("testdata_real", (0,))])
def test_gausslq_nbp_rows_regenerate(emu, name, rows):
    q = emu.load("gausslq")
    s, ref = golden("gausslq_" + name), golden("gausslq_" + name + "_nbp")
    for i in rows:
        spot = s["spots"][i]
        size = spot.shape[0]
        assert np.array_equal(q._initial_parameters(spot, size, int(size / 2)), ref["theta0"][i])
        assert np.array_equal(np.asarray(q.fit_spot(spot)), ref["theta"][i])


def test_emulator_rules():
    """The typing rules themselves, on scalars and arrays (tests/golden/_nbemu.py docstring)."""
    sys.path.insert(0, GOLDEN)
    import _nbemu as e
    f32, f64, i64 = np.float32, np.float64, np.int64
    assert type(e.binop("Sub", f32(1.5), 2)) is f64 and type(e.binop("Mult", f32(1.5), f32(2))) is f32
    assert type(e.binop("Add", f32(1.5), 0.25)) is f64 and type(e.binop("Div", 3, 2)) is f64
    p = e.binop("Pow", f32(1.1), 3)
    assert type(p) is f32 and p == f32(f32(1.1) * f32(f32(1.1) * f32(1.1)))
    assert type(e.binop("Pow", f32(2.0), -1)) is f32 and type(e.binop("Pow", i64(3), 2)) is i64
    a = np.arange(4, dtype=f32)
    assert e.binop("Sub", a, 1).dtype == f32                  # integers give way to the float type of the array
    assert e.binop("Mult", 0.1, a).dtype == f64               # no value-based casting of a float64 scalar
    assert e.binop("Sub", a, f32(1)).dtype == f32 and e.binop("Pow", a.astype(f64), 2).dtype == f64
    assert e.binop("Add", "a", "b") == "ab"
