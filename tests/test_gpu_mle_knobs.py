"""GPU tier: MLE parity at the settings the reference's GUI exposes, not only its defaults.

picasso/gui/localize.py:996-1008 lets the user set the convergence criterion `eps` and `max_it`; the CLI takes any
box.  Every case runs the default mode (float32 loop + re-fit of the flagged spots in the reference's arithmetic)
on adversarial spots — wide and narrow widths, centres 1.5 px off, 20 photons, negative pixels — and asserts, on
EVERY row (conftest.assert_mle_rows, no mask): the oracle's iteration count; where the oracle converged, x, y and
the widths within max(1e-3 px, eps) (a coarser eps than the north star's tolerance leaves the converged position
undetermined to eps) and photons within 1e-2.  Reference: picasso/gaussmle.py:632-638, :844-852 (the stop test),
:860-884 / :647-670 (the update).
"""
import numpy as np
import pytest

from conftest import assert_mle_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from picasso_amd import backend
    return backend


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def knob_spots(box, n, seed, negative=True):
    """The spots of tools/fuzz_parity.py's MLE branch, vectorised: widths 0.5 px ... 0.3 box + 0.5, centres up to
    1.5 px off, 20 ... 9000 photons on 0.05 ... 60 background photons, a third of them shifted down by 3 (negative
    pixels: a baseline set too high)."""
    rng = np.random.default_rng(seed)
    c, idx = box // 2, np.arange(box)
    x0 = c + rng.uniform(-1.5, 1.5, n)
    y0 = c + rng.uniform(-1.5, 1.5, n)
    sx = rng.uniform(0.5, 0.3 * box + 0.5, n)
    sy = rng.uniform(0.5, 0.3 * box + 0.5, n)
    gx = np.exp(-0.5 * ((idx[None, :] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
    gy = np.exp(-0.5 * ((idx[None, :] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
    lam = rng.uniform(20, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :] + rng.uniform(0.05, 60, n)[:, None, None]
    spots = rng.poisson(lam).astype(np.float32)
    if negative:
        spots -= rng.choice(np.float32([0.0, 0.0, 3.0]), n)[:, None, None]
    return spots


def check_case(be, orc, spots, eps, max_it, method, label):
    o = orc.gaussmle(spots, eps, max_it, method, threads=orc.max_threads())
    g = be.gaussmle_arrays(spots, eps, max_it, method)
    tol = max(1e-3, eps)
    assert_mle_rows(g[0][:, 0], g[0][:, 1], g[0][:, 4], g[0][:, 5], g[0][:, 2], g[3],
                    o[0][:, 0], o[0][:, 1], o[0][:, 4], o[0][:, 5], o[0][:, 2], o[3], max_it=max_it, label=label, tol_px=tol)
    return g, o


# eps x max_it at the boxes of configs 2 and 5 and at the largest box of each kernel family
KNOBS = [(1e-3, 100), (1e-2, 100), (1e-4, 100), (1e-3, 5), (1e-2, 5), (1e-4, 5)]


@pytest.mark.parametrize("box,n", [(3, 20000), (7, 20000), (13, 20000), (15, 20000), (17, 20000), (21, 20000)])
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
@pytest.mark.parametrize("eps,max_it", KNOBS)
def test_gaussmle_every_row_over_eps_max_it_boxes(be, orc, box, n, method, eps, max_it):
    spots = knob_spots(box, n, 1000 * box + int(-np.log10(eps)) * 10 + max_it)
    check_case(be, orc, spots, eps, max_it, method, f"box {box} {method} eps {eps} max_it {max_it}")


@pytest.mark.parametrize("box,n", [(3, 20000), (7, 20000), (13, 6000)])
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
@pytest.mark.parametrize("eps", [1e-2, 1e-3, 1e-4])
def test_gaussmle_every_row_at_max_it_1000(be, orc, box, n, method, eps):
    """max_it = 1000: the fits that never settle run ten times longer than at the default; the GUI allows 1e6."""
    spots = knob_spots(box, n, 77 * box + int(-np.log10(eps)))
    check_case(be, orc, spots, eps, 1000, method, f"box {box} {method} eps {eps} max_it 1000")


@pytest.mark.parametrize("box", [7, 13])
@pytest.mark.parametrize("max_it", [40, 64])
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_every_row_with_max_it_between_32_and_64(be, orc, box, max_it, method):
    """max_it in (32, 64] at eps 1e-4: fits that run into max_it, or take more than 32 iterations, carry no flag of their own
    since round 3 (the slow-fit threshold went from 32 to 64, the 'ran into max_it' flag was dropped) — the margin, swing and
    contraction flags have to cover them here too."""
    spots = knob_spots(box, 20000, 4000 + 100 * box + max_it)
    check_case(be, orc, spots, 1e-4, max_it, method, f"box {box} {method} eps 1e-4 max_it {max_it}")


def test_regression_swinging_fit_wide_sigma(be, orc):
    """fuzz_long.log (round 2): box 21 `sigma`, eps 1e-3: equal iteration counts (14 / 14) but 1.07e-3 px apart — photons,
    background and width trade against each other and step back and forth for a dozen iterations; the positions pass
    the test while the width is still swinging.  Such spots carry the `swing` flag now."""
    spots = knob_spots(21, 30000, 2021)
    g, o = check_case(be, orc, spots, 1e-3, 100, "sigma", "box 21 sigma, swinging fits")
    why = be.last_flag_reasons()
    assert why["swing"] > 0 and why["wild"] > 0, why
    spots = knob_spots(15, 30000, 2015)
    check_case(be, orc, spots, 1e-2, 100, "sigma", "box 15 sigma eps 1e-2")


def test_regression_small_max_it(be, orc):
    """fuzz_long.log: box 21 `sigmaxy` max_it 5: iterations 3 vs 4 (a model pinned at the 0.01 background floor under
    negative pixels: |data / model| in the thousands); and the re-fit must not take every spot that merely ran into a
    small max_it — at max_it = 5 most healthy fits do."""
    spots = knob_spots(21, 30000, 521)
    check_case(be, orc, spots, 1e-3, 5, "sigmaxy", "box 21 sigmaxy max_it 5")
    from math import erf, sqrt
    rng = np.random.default_rng(5)
    n, box, c = 20000, 7, 3
    idx = np.arange(box)
    real = np.empty((n, box, box), np.float32)
    x0 = c + rng.uniform(-0.6, 0.6, n); y0 = c + rng.uniform(-0.6, 0.6, n); s = rng.uniform(0.9, 1.4, n)
    ex = 0.5 * (np.vectorize(erf)((idx[None] - x0[:, None] + .5) / (sqrt(2) * s[:, None])) - np.vectorize(erf)((idx[None] - x0[:, None] - .5) / (sqrt(2) * s[:, None])))
    ey = 0.5 * (np.vectorize(erf)((idx[None] - y0[:, None] + .5) / (sqrt(2) * s[:, None])) - np.vectorize(erf)((idx[None] - y0[:, None] - .5) / (sqrt(2) * s[:, None])))
    real[:] = rng.poisson(rng.uniform(2000, 8000, n)[:, None, None] * ey[:, :, None] * ex[:, None, :] + rng.uniform(10, 30, n)[:, None, None])
    check_case(be, orc, real, 1e-3, 5, "sigmaxy", "real-like 7x7 max_it 5")
    assert be.last_refit_count() < 0.05 * n, be.last_refit_count()


def test_regression_tiny_boxes(be, orc):
    """fuzz_long.log: two 3x3 `sigma` batches with an iteration mismatch (29 / 29 and 7 / 7 on the rows shown, another row
    of the batch one apart)."""
    for seed in (3, 33, 333):
        spots = knob_spots(3, 30000, seed)
        check_case(be, orc, spots, 1e-3, 100, "sigma", f"box 3 sigma seed {seed}")


def test_fuzz_residual_unstable_iteration_box13(be, orc):
    """The 13x13 `sigmaxy` fit of round 3's fuzz run that no step-sequence rule caught (14 iterations on the device, 16 in the
    reference, 2e-3 px apart): lambda_max of its normalised Fisher matrix is 2.356 — the differences between the two
    arithmetics alternate with a factor of -1.35 — and the Fisher pass now sends such spots to a second re-fit."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "mle_fuzz_regressions", "mle_31_2711.npz"))
    spots = np.repeat(z["spots"], 64, axis=0)
    check_case(be, orc, spots, float(z["eps"]), int(z["max_it"]), str(z["method"]), "fuzz residual box 13")
    assert be.last_flag_reasons()["unstable"] == 64


def _fuzz_residual_files():
    import glob
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mle_fuzz_regressions")
    return sorted(glob.glob(os.path.join(here, "mle_*.npz")))


@pytest.mark.parametrize("path", _fuzz_residual_files(), ids=lambda p: p.rsplit("/", 1)[-1][:-4])
def test_fuzz_residuals_are_bounded(be, orc, path):
    """Every spot tools/fuzz_parity.py ever left (tests/golden/mle_fuzz_regressions, 75 million spots over rounds 3 and 4),
    one test each: the fit must be the oracle's on every row.  The 13x13 spot of round 3 has been since its `unstable` flag;
    the sixteen 3x3 `sigmaxy` fits whose width collapses to ~0.03 px on one axis (87 ... 342 iterations, a chaotic trajectory)
    since round 6: the reference-arithmetic kernel evaluates erf / exp with the bits of the reference's C library
    (csrc/libm_glibc.h) — with the device library's functions they ended up to 0.158 px and 116 iterations from the oracle,
    which the second half of the test keeps on record (PMI_LIBM_DEVICE: bounded by what each file recorded).
    Reference: picasso/gaussmle.py:745-857 (math.erf at :279)."""
    z = np.load(path)
    spots, eps, max_it, method = z["spots"], float(z["eps"]), int(z["max_it"]), str(z["method"])
    check_case(be, orc, spots, eps, max_it, method, path)
    if int(z["box"]) >= 5:
        return
    assert method == "sigmaxy" and eps <= 1e-3
    o = orc.gaussmle(spots, eps, max_it, method, threads=1)
    be.set_mle_libm("device")
    try:
        g = be.gaussmle_arrays(spots, eps, max_it, method)
    finally:
        be.set_mle_libm("auto")
    was = np.abs(z["theta_gpu"].astype(np.float64) - z["theta_orc"])[0]
    now = np.abs(g[0].astype(np.float64) - o[0])[0]
    was_it = abs(int(z["it_gpu"][0]) - int(z["it_orc"][0]))
    assert abs(int(g[3][0]) - int(o[3][0])) <= was_it + 2, (int(g[3][0]), int(o[3][0]), was_it)
    collapsed = int(np.argmin(o[0][0, 4:6]))                 # 0: the x axis carries the collapsed width, 1: the y axis
    assert o[0][0, 4 + collapsed] < 0.04, o[0][0]
    for col in (0, 1, 4, 5):
        if col % 2 == collapsed:                             # x, sx (or y, sy) of the collapsed axis
            assert now[col] <= 1.25 * was[col] + 1e-4, (col, now[col], was[col])
        else:
            assert now[col] <= 1e-3 * max(1.0, abs(o[0][0, col])), (col, now[col])
    assert now[2] <= 1e-2 * abs(o[0][0, 2]) and now[3] <= 1e-2 * max(1.0, abs(o[0][0, 3]))


def _round6_files():
    import glob
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mle_fuzz_regressions")
    return sorted(glob.glob(os.path.join(here, "r6_*.npz")))


@pytest.mark.parametrize("path", _round6_files(), ids=lambda p: p.rsplit("/", 1)[-1][:-4])
def test_round6_fuzz_finds_carry_the_oracles_bits(be, orc, path):
    """What tools/fuzz_mle_libm.py (narrow, off-centre spots in 3x3 ... 9x9 boxes, max_it up to 1000) found in round 6, each spot
    in both modes — `strict`: theta and iterations bit for bit; default: iterations, 1e-3 px:
      r6_libm_strict_*: 3x3 `sigma` fits whose centre sits on the first pixel's middle to 1e-17 px — (k - mu) + 1/2 and
          (k + 1 - mu) - 1/2 then round differently, the kernel's shared boundary records were not the reference's per-pixel
          values (the coordinate came out 1e-17 px off; `split` in gaussmle_strict.hip);
      r6_exploded_*: 7x7 / 9x9 `sigmaxy` fits that wander for max_it = 1000 iterations with widths of hundreds of pixels: the
          device library's erf / exp flip a float32 rounding about once per such fit (csrc/libm_glibc.h);
      r6_libm_refit_*, r6_box7_*: 5x5 / 7x7 fits of 0.31 ... 0.495 px width at eps 1e-4 that ended one or more iterations off
          the reference without raising a flag (FIT_NARROW_SIGMA 0.3 -> 0.5).
    Reference: picasso/gaussmle.py:268-303, 745-857."""
    z = np.load(path)
    spots, eps, max_it, method = z["spots"], float(z["eps"]), int(z["max_it"]), str(z["method"])
    o = orc.gaussmle(spots, eps, max_it, method, threads=1)
    be.set_mle_mode("strict")
    try:
        g = be.gaussmle_arrays(spots, eps, max_it, method)
    finally:
        be.set_mle_mode("refit")
    assert np.array_equal(g[3], o[3]) and np.array_equal(g[0], o[0], equal_nan=True), (g[3], o[3], g[0], o[0])
    g = be.gaussmle_arrays(spots, eps, max_it, method)
    assert np.array_equal(g[3], o[3]), (g[3], o[3])
    fin = np.all(np.isfinite(o[0]), axis=1) & (o[3] < max_it)
    assert np.all(np.abs(g[0][fin][:, [0, 1, 4, 5]] - o[0][fin][:, [0, 1, 4, 5]]) <= 1e-3)


def test_open_residual_runaway_fit_with_a_cancelling_curvature_term(be, orc):
    """The one spot the round's last fuzz run left (tests/golden/mle_fuzz_regressions/open): a 7x7 `sigmaxy` fit
    whose curvature term for sigma_x cancels to 2e-6 of its two sums at the start values — positive in the reference, which then
    runs away to max_it under its step clamp; negative in the float32 loop, which converges, and no flag asks.  The strict mode
    must be the reference's bits; the default mode is recorded as an expected failure until the curvature flag looks at the
    size of the term and not only at its sign."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "mle_fuzz_regressions", "open", "mle_6601_1014.npz"))
    spots, eps, max_it, method = z["spots"], float(z["eps"]), int(z["max_it"]), str(z["method"])
    o = orc.gaussmle(spots, eps, max_it, method, threads=1)
    be.set_mle_mode("strict")
    try:
        g = be.gaussmle_arrays(spots, eps, max_it, method)
    finally:
        be.set_mle_mode("refit")
    assert np.array_equal(g[3], o[3]) and np.array_equal(g[0], o[0], equal_nan=True)
    g = be.gaussmle_arrays(spots, eps, max_it, method)
    if not np.array_equal(g[3], o[3]):
        pytest.xfail(f"open residual (DESIGN.md 10.8): {int(g[3][0])} iterations against the reference's {int(o[3][0])}")


@pytest.mark.parametrize("box,n,groups", [(7, 120000, 12288), (13, 40000, 6144)])
def test_refit_lists_longer_than_one_round_of_the_refit_kernel(be, orc, box, n, groups):
    """The re-fit kernel deals the first `groups` entries of its list to its lane groups and hands the rest out through a
    queue word (csrc/gaussmle_strict.hip).  At eps 1e-4 the margin flag sends a good share of these spots there: the list
    is several rounds long, and every row must still be the oracle's."""
    spots = knob_spots(box, n, 4242 + box)
    check_case(be, orc, spots, 1e-4, 100, "sigmaxy", f"box {box}, {n} spots, eps 1e-4")
    assert be.last_refit_count() > 1.5 * groups, be.last_refit_count()
