"""CPU tier: the C oracle against the golden vectors minted from the reference.

These tests pin the oracle (oracle/picasso_oracle.c).  They never touch the
product package.
"""
import numpy as np
import pytest

from conftest import DEGENERATE_LOOSE, MLE_DATASETS, bounds_from, golden, roi_from
from oracle import oracle as orc


def _sorted(g, k):
    o = np.lexsort((g[k + "_x"], g[k + "_y"], g[k + "_frame"]))
    return g[k + "_frame"][o], g[k + "_y"][o], g[k + "_x"][o], g[k + "_ng"][o]


@pytest.mark.parametrize("case", list("abcdefgh"))
def test_identify_testdata_bit_exact(testdata_movie, case):
    g = golden("identify_testdata")
    fr, y, x, ng = orc.identify(testdata_movie, float(g[case + "_min_ng"]), int(g[case + "_box"]),
                                roi_from(g[case + "_roi"]), bounds_from(g[case + "_frame_bounds"]))
    gf, gy, gx, gn = _sorted(g, case)
    assert len(fr) == len(gf)
    assert np.array_equal(fr, gf) and np.array_equal(y, gy) and np.array_equal(x, gx)
    assert np.array_equal(ng, gn)  # float32 accumulation order is part of the contract


@pytest.mark.parametrize("case", list("abcd"))
def test_identify_adversarial_bit_exact(case):
    """Plateaus (first-argmax tie rule), the y==h / x==h wrap band, ROI."""
    g = golden("identify_adversarial")
    fr, y, x, ng = orc.identify(g["movie"], float(g[case + "_min_ng"]), int(g[case + "_box"]),
                                roi_from(g[case + "_roi"]), None)
    gf, gy, gx, gn = _sorted(g, case)
    assert np.array_equal(fr, gf) and np.array_equal(y, gy) and np.array_equal(x, gx)
    assert np.array_equal(ng, gn)
    if case == "a":  # the forced wrap-band candidates must be present
        assert ((y == 3) & (x == 3)).any() and ((y == 20) & (x == 3)).any()


def test_identify_matches_real_numba_table(testdata_movie):
    """frame / net_gradient written by a real numba Picasso run (bundled HDF5)."""
    nb = golden("numba_identifications_testdata")
    fr, y, x, ng = orc.identify(testdata_movie, 5000, 7)
    assert np.array_equal(fr, nb["frame"])
    assert np.array_equal(ng, nb["net_gradient"])
    assert np.all(np.abs(x - nb["x_fit"]) < 1.0) and np.all(np.abs(y - nb["y_fit"]) < 1.0)


def test_identify_threads_equal_serial(testdata_movie):
    a = orc.identify(testdata_movie, 400, 7, threads=1)
    b = orc.identify(testdata_movie, 400, 7, threads=4)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_get_spots_bit_exact(testdata_movie):
    s = golden("get_spots_testdata")
    for key in ("unit", "emccd", "scmos"):
        cam = dict(zip(("Baseline", "Sensitivity", "Gain"), s["cam_" + key]))
        sp = orc.get_spots(testdata_movie, s["frame"], s["y"], s["x"], 7, cam)
        assert sp.dtype == np.float32 and sp.shape == (30, 7, 7)
        assert np.array_equal(sp, s["spots_" + key])
    sp = orc.get_spots(testdata_movie, s["box9_frame"], s["box9_y"], s["box9_x"], 9,
                       {"Baseline": 0, "Sensitivity": 1, "Gain": 1})
    assert np.array_equal(sp, s["box9_spots"])


def _check_mle(d, prefix, th, cr, ll, it, loose=(), flip_frac=0.02, max_flip=1):
    gth, gcr, gll, git = (d[prefix + "_theta"], d[prefix + "_crlb"], d[prefix + "_loglik"],
                          d[prefix + "_iterations"])
    # numba-vs-NumPy promotion may flip a borderline |step| < eps test: +-1 iteration on
    # at most 2 % of spots at eps=1e-3, more at eps=1e-5 where a step is ~40 ulp;
    # flipped rows are excluded from the value comparison.
    assert np.max(np.abs(it.astype(int) - git.astype(int))) <= max_flip
    assert np.mean(it != git) <= flip_frac
    keep = np.array([i not in loose for i in range(len(it))]) & (it == git)
    # oracle = numba promotion, goldens = NumPy promotion: <= 1e-4 px; observed 5e-7
    assert np.nanmax(np.abs(th[keep, :2] - gth[keep, :2])) < 1e-4
    assert np.nanmax(np.abs(th[keep, 4:] - gth[keep, 4:])) < 1e-4
    rel = np.abs(th[keep, 2] - gth[keep, 2]) / np.maximum(np.abs(gth[keep, 2]), 1.0)
    assert np.nanmax(rel) < 1e-5
    assert np.nanmax(np.abs(th[keep, 3] - gth[keep, 3])) < 1e-3
    with np.errstate(invalid="ignore"):
        lp, glp = np.sqrt(cr[keep]), np.sqrt(gcr[keep])
    ok = np.isfinite(glp)
    assert np.array_equal(np.isfinite(lp), ok)
    assert np.nanmax(np.abs(lp[ok] - glp[ok]) / np.maximum(glp[ok], 1e-6)) < 1e-3
    assert np.nanmax(np.abs(ll[keep] - gll[keep])) < 2e-2


@pytest.mark.parametrize("name", MLE_DATASETS)
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_against_goldens(name, method):
    d = golden("gaussmle_" + name)
    th, cr, ll, it = orc.gaussmle(d["spots"], 1e-3, 100, method)
    assert th.dtype == np.float32 and th.shape == (len(d["spots"]), 6)
    assert it.dtype == np.int32 and ll.dtype == np.float32
    loose = DEGENERATE_LOOSE if name == "degenerate7" else ()
    _check_mle(d, method, th, cr, ll, it, loose)
    if method == "sigma":
        assert np.array_equal(th[:, 4], th[:, 5])  # reference test_gaussmle.py:72-76


@pytest.mark.parametrize("name", ["conftest_noisy", "poisson7"])
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_iteration_limited_and_tight_eps(name, method):
    d = golden("gaussmle_" + name)
    th, cr, ll, it = orc.gaussmle(d["spots"], 1e-3, 3, method)
    _check_mle(d, method + "_it3", th, cr, ll, it)
    assert it.max() <= 3
    th, cr, ll, it = orc.gaussmle(d["spots"], 1e-5, 100, method)
    _check_mle(d, method + "_eps5", th, cr, ll, it, flip_frac=0.10, max_flip=3)


# ---------------------------------------------------------------------------
# The pin under NUMBA's type promotion: tests/golden/*_nbp.npz are the reference's own gaussmle.py / gausslq.py executed
# with the arithmetic of their @numba.jit functions typed by numba's rules (tests/golden/_nbemu.py; minted under NumPy
# 1.26.4, whose scalar promotion is numba's, reproduced bit for bit under NumPy 2.2 + SciPy 1.15.3).  No tolerance, no
# loose row, degenerate7 included: theta, the iteration count and the log-likelihood are EQUAL on every row.
# ---------------------------------------------------------------------------
NBP_VARIANTS = [("", 1e-3, 100), ("_it3", 1e-3, 3), ("_eps5", 1e-5, 100)]


def _rows_equal(a, b):
    return np.array([np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b)])


@pytest.mark.parametrize("name", MLE_DATASETS)
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_equals_numba_promotion_goldens(name, method):
    d, g = golden("gaussmle_" + name), golden("gaussmle_" + name + "_nbp")
    for tag, eps, max_it in NBP_VARIANTS:
        key = method + tag
        if key + "_theta" not in g.files:
            assert tag, "the default variant is minted for every dataset"
            continue
        th, cr, ll, it = orc.gaussmle(d["spots"], eps, max_it, method)
        assert np.array_equal(it, g[key + "_iterations"]), f"{name} {key}: iterations"
        bad = np.flatnonzero(~_rows_equal(th, g[key + "_theta"]))
        assert len(bad) == 0, f"{name} {key}: theta differs on rows {bad[:8]}"
        bad = np.flatnonzero(~_rows_equal(ll, g[key + "_loglik"]))
        assert len(bad) == 0, f"{name} {key}: log-likelihood differs on rows {bad[:8]}"
        # CRLB = diag(pinv(M)): third-party arithmetic (LAPACK gesdd in NumPy and in numba, a Jacobi eigen-solver here).
        # Well-posed Fisher matrices: within one float32 ulp (observed: equal on 99 % of rows, <= 1.1e-7 relative).  A
        # (near-)singular matrix (degenerate7: flat / single-pixel / unconverged spots) leaves the cut-off singular
        # values to the solver: there the bound is 1e-5 of the row's largest entry.
        ref = g[key + "_crlb"]
        with np.errstate(invalid="ignore"):
            diff = np.abs(cr.astype(np.float64) - ref)
        diff[(cr == ref) | (np.isnan(cr) & np.isnan(ref))] = 0.0
        assert np.array_equal(np.isinf(cr), np.isinf(ref))
        tol = 2e-7 * np.abs(ref)
        if name == "degenerate7":
            tol = np.maximum(tol, 1e-5 * np.nanmax(np.abs(ref), axis=1, keepdims=True))
        assert np.all(diff <= tol), f"{name} {key}: CRLB"
    if name == "degenerate7":
        # rows on which the production path (numba, error_model="python") raises ZeroDivisionError in
        # _initial_sigmas (gaussmle.py:113-114, a centre row / column summing to zero after the background is
        # removed): IEEE semantics are followed there, here and in the oracle; recorded, not hidden
        assert np.flatnonzero(g[method + "_zero_division"]).tolist() == [0, 1, 3]
    else:
        assert not g[method + "_zero_division"].any()


@pytest.mark.parametrize("name", ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "poisson13"])
def test_gausslq_equals_numba_promotion_goldens(name):
    """Start values (gausslq.py:52-112, float64 moment sums under numba) and the fit FROM THE ORACLE'S OWN start
    (scipy's MINPACK lmdif over the reference's residual function typed by numba's rules): equal, every row."""
    s, g = golden("gausslq_" + name), golden("gausslq_" + name + "_nbp")
    assert np.array_equal(orc.gausslq_initial(s["spots"]), g["theta0"])
    assert np.array_equal(orc.gausslq(s["spots"], threads=2), g["theta"].astype(np.float32))      # fit_spots stores float32 (gausslq.py:275)
    # and from the NumPy-2 goldens' start values the numba-typed residuals lead MINPACK to the NumPy-2 goldens' result
    assert np.array_equal(g["theta_from_golden0"].astype(np.float32), s["theta"])


def test_gaussmle_ground_truth_recovery():
    """The reference's own tolerance test (tests/test_gaussmle.py:50-70)."""
    box, n = 7, 64
    rng = np.random.default_rng(42)
    gt = {k: rng.uniform(*r, n) for k, r in (("x", (-0.5, 0.5)), ("y", (-0.5, 0.5)), ("sx", (0.9, 1.4)),
                                            ("sy", (0.9, 1.4)), ("photons", (2000.0, 8000.0)),
                                            ("bg", (5.0, 30.0)))}
    d = golden("gaussmle_conftest_clean")
    th, cr, ll, it = orc.gaussmle(d["spots"], 1e-3, 100, "sigmaxy")
    assert np.all(np.abs(th[:, 0] - box // 2 - gt["x"]) < 0.05)
    assert np.all(np.abs(th[:, 1] - box // 2 - gt["y"]) < 0.05)
    assert np.all(np.abs(th[:, 2] - gt["photons"]) / gt["photons"] < 0.05)
    assert np.all(np.isfinite(cr)) and np.all(cr > 0)


def test_gaussmle_threads_equal_serial():
    d = golden("gaussmle_poisson7")
    a = orc.gaussmle(d["spots"], 1e-3, 100, "sigmaxy", threads=1)
    b = orc.gaussmle(d["spots"], 1e-3, 100, "sigmaxy", threads=4)
    assert all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b))


def test_gaussmle_bad_method_and_max_it_zero():
    d = golden("gaussmle_poisson5")
    with pytest.raises(ValueError, match="Method not available"):
        orc.gaussmle(d["spots"], 1e-3, 100, "nope")
    th, cr, ll, it = orc.gaussmle(d["spots"][:4], 1e-3, 0, "sigmaxy")
    assert np.all(it == 0)
    assert np.allclose(th, orc.initial_parameters(d["spots"][:4]))


def test_zfit_against_goldens():
    """Bounded Brent restatement vs the reference running scipy's minimize_scalar."""
    g = golden("zfit_calib3d")
    z, sq = orc.zfit(g["sx"], g["sy"], g["cx"], g["cy"], threads=2)
    zz = z.astype(np.float32) * np.float32(g["magnification"])
    dz = np.sqrt(sq.astype(np.float32))
    for m in ("gausslq", "gaussmle"):
        idx = g[m + "_index"]
        # the NumPy-executed reference takes sqrt(sx) in float32 (numba: float64): <= 1 float32 ulp in z
        assert np.max(np.abs(zz[idx] - g[m + "_z"])) <= 1.3e-4
        assert np.max(np.abs(dz[idx] - g[m + "_d_zcalib"])) < 1e-6


def test_avgroi_bit_exact():
    a = golden("avg_testdata")
    s = golden("get_spots_testdata")
    assert np.array_equal(orc.avgroi(s["spots_unit"]), a["theta"])


@pytest.mark.parametrize("name", ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "poisson13"])
def test_gausslq_against_goldens(name):
    """MINPACK lmdif restatement + the point-sampled Gaussian residuals (gausslq.py:33-244).
    From the reference's own start values the result is bit-identical to what the reference got
    out of scipy.optimize.leastsq; from the oracle's start values (float64 moment sums, numba
    promotion) it differs by what ftol = xtol = 1e-2 leaves undetermined."""
    g = golden("gausslq_" + name)
    th = orc.gausslq_from(g["spots"], g["theta0"], threads=2)
    assert np.array_equal(th, g["theta"])
    t0 = orc.gausslq_initial(g["spots"])
    assert np.all(np.abs(t0 - g["theta0"]) <= 3e-6 + 1e-6 * np.abs(g["theta0"]))     # float32-vs-float64 moment sums
    full = orc.gausslq(g["spots"], threads=2)
    assert np.max(np.abs(full[:, :2] - g["theta"][:, :2])) < 5e-3
    assert np.max(np.abs(full[:, 4:] - g["theta"][:, 4:])) < 5e-3
    assert np.max(np.abs(full[:, 2] - g["theta"][:, 2]) / g["theta"][:, 2]) < 5e-3


def test_gausslq_matches_scipy_minpack():
    """The lmdif restatement against scipy's MINPACK on the same residual function."""
    from scipy import optimize

    def resid(theta, spot, size):
        h = size // 2
        grid = np.arange(-h, h + 1, dtype=np.float32).astype(np.float64)
        mx = (0.3989422804014327 / theta[4] * np.exp(-0.5 * ((grid - theta[0]) / theta[4]) ** 2)).astype(np.float32)
        my = (0.3989422804014327 / theta[5] * np.exp(-0.5 * ((grid - theta[1]) / theta[5]) ** 2)).astype(np.float32)
        model = (theta[2] * my.astype(np.float64)[:, None] * mx.astype(np.float64)[None, :] + theta[3]).astype(np.float32)
        return (spot - model).astype(np.float32).ravel()

    g = golden("gausslq_poisson7")
    spots = g["spots"][:48]
    t0 = orc.gausslq_initial(spots)
    th, info, nfev = orc.gausslq(spots, full=True)
    for i in range(len(spots)):
        r = optimize.leastsq(resid, t0[i], args=(spots[i], 7), ftol=1e-2, xtol=1e-2, full_output=True)
        assert np.array_equal(r[0].astype(np.float32), th[i]) and r[4] == info[i] and r[2]["nfev"] == nfev[i]


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_render_against_goldens(case):
    """render._render_hist / _render_gaussian (picasso/render.py:177-232, 451-467, 494-575).
    'numba' goldens: the reference's own functions fed float64-widened inputs, i.e. numba's promotion
    of the float32 columns -> bit-identical.  'numpy' goldens: the reference as NumPy executes it
    (float32 coordinates), which moves narrow Gaussians by ~1e-6 px."""
    g = golden("render_cases")
    vp = [tuple(g[case + "_viewport"][0]), tuple(g[case + "_viewport"][1])]
    osamp, mbw = float(g[case + "_oversampling"]), float(g[case + "_min_blur"])
    n, hist = orc.render(g["x"], g["y"], osamp, vp)
    assert n == int(g[case + "_n"]) and np.array_equal(hist, g[case + "_hist"])
    n, img = orc.render(g["x"], g["y"], osamp, vp, g["lpx"], g["lpy"], "gaussian", mbw)
    assert n == int(g[case + "_n"]) and img.dtype == np.float32
    assert np.array_equal(img, g[case + "_gauss_numba"])
    ref = g[case + "_gauss_numpy"]
    assert np.max(np.abs(img - ref)) < 2e-4 * max(1.0, float(ref.max()))
    assert abs(float(img.sum()) - float(ref.sum())) < 1e-3 * float(ref.sum())
    n, iso = orc.render(g["x"], g["y"], osamp, vp, g["lpx"], g["lpy"], "gaussian_iso", mbw)
    assert n == int(g[case + "_n"]) and np.array_equal(iso, g[case + "_iso_numba"])
    assert np.max(np.abs(iso - g[case + "_iso_numpy"])) < 2e-4 * max(1.0, float(g[case + "_iso_numpy"].max()))


@pytest.mark.parametrize("tag,box", [("ng7", 7), ("ng5", 5), ("ng9", 9)])
def test_oracle_net_gradient_standalone(tag, box):
    """orc_net_gradient against the reference's _net_gradient on single frames (local maxima plus positions whose
    window wraps through row / column -1), with the standard and with an arbitrary unit-vector table: bit for bit."""
    from oracle import oracle as orc
    g = golden("surface_cases")
    frame, y, x = g[f"{tag}_frame"], g[f"{tag}_y"], g[f"{tag}_x"]
    assert np.array_equal(orc.net_gradient(frame, y, x, box, g[f"{tag}_uy"], g[f"{tag}_ux"]), g[f"{tag}_ng"])
    assert np.array_equal(orc.net_gradient(frame, y, x, box, g[f"{tag}_ruy"], g[f"{tag}_rux"]), g[f"{tag}_rng"])
    ux, uy = orc.unit_vectors(box)
    c = box // 2
    keep = np.ones((box, box), bool); keep[c, c] = False
    assert np.array_equal(ux[keep], g[f"{tag}_ux"][keep]) and np.array_equal(uy[keep], g[f"{tag}_uy"][keep])
    with pytest.raises(ValueError):
        orc.net_gradient(frame, [31 - box // 2], [10], box, uy, ux)


# ---------------------------------------------------------------------------
# RCC peak fit: the oracle's Trust Region Reflective restatement against scipy itself
# ---------------------------------------------------------------------------
def _peak_windows(n, box, seed):
    rng = np.random.default_rng(seed)
    h = box // 2
    y, x = np.mgrid[-h:h + 1, -h:h + 1]
    out = []
    for t in range(n):
        a = rng.uniform(5, 500); xc, yc = rng.uniform(-0.7, 0.7, 2); s = rng.uniform(0.6, 2.5)
        b = rng.uniform(0, 50) if t % 3 else 0.0                      # every third window: background on its bound
        roi = a * np.exp(-0.5 * ((x - xc) ** 2 + (y - yc) ** 2) / s ** 2) + b + rng.normal(0, 0.02 * a, (box, box))
        out.append(np.abs(roi) if t % 2 else np.maximum(roi, 0.0))
    return np.array(out)


@pytest.mark.parametrize("box", [5, 7])
def test_peak_fit_matches_scipy_curve_fit(box):
    """orc_peak_fit restates scipy.optimize.curve_fit(bounds=...) = least_squares(method="trf") — third-party
    arithmetic (scipy 1.15.3), call site picasso/imageprocess.py:129-135.  Same start, same bounds, same
    tolerances: the fitted centre agrees with scipy's to 1e-7 px on every window, the termination status too."""
    from scipy.optimize import curve_fit
    from oracle import oracle as orc
    h = box // 2
    y, x = np.mgrid[-h:h + 1, -h:h + 1]

    def flat(coords, a, xc, yc, s, b):
        xx, yy = coords
        return (a * np.exp(-0.5 * ((xx - xc) ** 2 + (yy - yc) ** 2) / s ** 2) + b).flatten()

    rois = _peak_windows(120, box, 7 + box)
    for roi in rois:
        popt, _ = curve_fit(flat, (x, y), roi.flatten(), p0=[roi.max(), 0, 0, 1, roi.min()],
                            bounds=([0, -np.inf, -np.inf, 0, 0], [np.inf] * 5))
        got, status, nfev = orc.peak_fit(roi)
        assert status in (1, 2, 3, 4) and nfev < 500
        assert np.max(np.abs(got[1:3] - popt[1:3])) < 1e-7
        assert abs(got[3] - popt[3]) < 1e-6 * max(1.0, popt[3]) and abs(got[0] - popt[0]) < 1e-6 * popt[0]
    with pytest.raises(ValueError, match="outside of provided bounds"):
        orc.peak_fit(rois[0] - rois[0].max())                         # curve_fit refuses b0 < 0
