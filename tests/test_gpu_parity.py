"""GPU tier: the HIP path (through the C ABI) against the oracle and the goldens.

Run on the MI355X box: python -m pytest tests -m gpu -x -q
Tolerances (BASELINE.json north_star): x, y, sigma within 1e-3 px, photons
within 1e-2 (relative 1e-5 here, tighter), identical identification set.
Integer/index work and the float32 net gradient are bit-exact.
"""
import numpy as np
import pytest

from conftest import DEGENERATE_LOOSE, MLE_DATASETS, assert_mle_rows, bounds_from, golden, roi_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from picasso_amd import backend
    return backend


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _sorted(g, k):
    o = np.lexsort((g[k + "_x"], g[k + "_y"], g[k + "_frame"]))
    return g[k + "_frame"][o], g[k + "_y"][o], g[k + "_x"][o], g[k + "_ng"][o]


@pytest.mark.parametrize("case", list("abcdefgh"))
def test_identify_testdata_bit_exact(be, testdata_movie, case):
    g = golden("identify_testdata")
    fr, y, x, ng = be.identify_arrays(testdata_movie, float(g[case + "_min_ng"]), int(g[case + "_box"]),
                                      roi_from(g[case + "_roi"]), bounds_from(g[case + "_frame_bounds"]))
    gf, gy, gx, gn = _sorted(g, case)
    assert len(fr) == len(gf)
    assert np.array_equal(fr, gf) and np.array_equal(y, gy) and np.array_equal(x, gx)
    assert np.array_equal(ng, gn)


@pytest.mark.parametrize("case", list("abcd"))
def test_identify_adversarial_bit_exact(be, case):
    g = golden("identify_adversarial")
    fr, y, x, ng = be.identify_arrays(g["movie"], float(g[case + "_min_ng"]), int(g[case + "_box"]),
                                      roi_from(g[case + "_roi"]), None)
    gf, gy, gx, gn = _sorted(g, case)
    assert np.array_equal(fr, gf) and np.array_equal(y, gy) and np.array_equal(x, gx)
    assert np.array_equal(ng, gn)


def test_identify_matches_real_numba_table(be, testdata_movie):
    nb = golden("numba_identifications_testdata")
    fr, y, x, ng = be.identify_arrays(testdata_movie, 5000, 7)
    assert np.array_equal(fr, nb["frame"]) and np.array_equal(ng, nb["net_gradient"])


@pytest.mark.parametrize("dtype", ["uint16", "uint8", "int16", "uint32", "int32", "float32"])
@pytest.mark.parametrize("shape,box", [((5, 67, 131), 7), ((3, 200, 300), 9), ((4, 40, 33), 5), ((2, 257, 129), 13)])
def test_identify_random_movies_vs_oracle(be, orc, dtype, shape, box):
    """Ragged shapes (tile edges), every pixel type, ties everywhere."""
    rng = np.random.default_rng(hash((dtype, shape, box)) % 2**32)
    hi = 200 if dtype == "uint8" else 3000
    mov = rng.integers(0, hi, size=shape).astype(dtype)
    mov[:, ::7, ::5] = mov[:, ::7, ::5] // 2 * 2    # more ties
    for min_ng in (-1e9, 0.3 * hi * box):
        a = be.identify_arrays(mov, min_ng, box)
        b = orc.identify(mov, min_ng, box, threads=4)
        assert len(a[0]) == len(b[0])
        assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_identify_empty_and_tiny(be):
    z = np.zeros((3, 16, 16), np.uint16)
    fr, y, x, ng = be.identify_arrays(z, 100, 7)
    assert len(fr) == 0
    fr, y, x, ng = be.identify_arrays(np.ones((2, 7, 7), np.uint16), -1, 7)   # no interior pixel
    assert len(fr) == 0
    fr, y, x, ng = be.identify_arrays(np.zeros((0, 32, 32), np.uint16), 1, 7)
    assert len(fr) == 0


def test_get_spots_bit_exact(be, testdata_movie):
    s = golden("get_spots_testdata")
    for key in ("unit", "emccd", "scmos"):
        b, se, g = s["cam_" + key]
        sp = be.get_spots_array(testdata_movie, s["frame"], s["y"], s["x"], 7, b, se, g)
        assert np.array_equal(sp, s["spots_" + key])
    sp = be.get_spots_array(testdata_movie, s["box9_frame"], s["box9_y"], s["box9_x"], 9, 0, 1, 1)
    assert np.array_equal(sp, s["box9_spots"])


def _check_fit(th, cr, ll, it, gth, gcr, gll, git, loose=(), max_it=100):
    """Every row (none masked by iteration count): identical iteration counts, north-star tolerance (tighter here)
    on all rows the reference converged on; `loose` rows (chaotic by construction) are compared on iterations only."""
    strict = np.array([i not in loose for i in range(len(it))])
    assert_mle_rows(th[strict, 0], th[strict, 1], th[strict, 4], th[strict, 5], th[strict, 2], it[strict],
                    gth[strict, 0], gth[strict, 1], gth[strict, 4], gth[strict, 5], gth[strict, 2], git[strict], max_it)
    assert np.all(np.abs(it.astype(int) - git.astype(int))[~strict] <= 1)
    keep = strict & (git < max_it)
    if not keep.any():
        return
    rel = np.abs(th[keep, 2] - gth[keep, 2]) / np.maximum(np.abs(gth[keep, 2]), 1.0)
    assert np.nanmax(rel) < 1e-4                                            # photons
    assert np.nanmax(np.abs(th[keep, 2] - gth[keep, 2])) < 0.25
    assert np.nanmax(np.abs(th[keep, 3] - gth[keep, 3])) < 1e-2            # background
    with np.errstate(invalid="ignore"):
        lp, glp = np.sqrt(cr[keep]), np.sqrt(gcr[keep])
    ok = np.isfinite(glp) & (glp > 0)
    assert np.nanmax(np.abs(lp[ok] - glp[ok]) / glp[ok]) < 2e-3
    assert np.nanmax(np.abs(ll[keep] - gll[keep])) < 0.05 + 2e-5 * np.nanmax(np.abs(gll[keep]))


@pytest.mark.parametrize("name", MLE_DATASETS)
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_vs_goldens_and_oracle(be, orc, name, method):
    d = golden("gaussmle_" + name)
    th, cr, ll, it = be.gaussmle_arrays(d["spots"], 1e-3, 100, method)
    assert th.dtype == np.float32 and th.shape == (len(d["spots"]), 6) and it.dtype == np.int32
    loose = DEGENERATE_LOOSE | {6, 7, 10, 11} if name == "degenerate7" else ()
    _check_fit(th, cr, ll, it, d[method + "_theta"], d[method + "_crlb"], d[method + "_loglik"],
               d[method + "_iterations"], loose)
    o = orc.gaussmle(d["spots"], 1e-3, 100, method, threads=4)
    # against the oracle (numba's promotion, which the device follows) NO row is loose:
    # the chaotic rows are flagged and carry the oracle's bits
    _check_fit(th, cr, ll, it, *o, ())
    if method == "sigma":
        assert np.array_equal(th[:, 4], th[:, 5])


NBP_VARIANTS = [("", 1e-3, 100), ("_it3", 1e-3, 3), ("_eps5", 1e-5, 100)]


@pytest.mark.parametrize("name", MLE_DATASETS)
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_vs_numba_promotion_goldens(be, name, method):
    """The device against the reference ITSELF under numba's typing (tests/golden/*_nbp.npz: the reference's gaussmle.py
    executed with its jitted arithmetic typed by numba's rules, tests/golden/_nbemu.py) — no oracle in between and no loose
    row: `strict` = theta and iterations bit for bit on every row (degenerate7 included), the default mode = the same
    iteration count on every row and the north-star tolerance wherever the reference converged."""
    d, g = golden("gaussmle_" + name), golden("gaussmle_" + name + "_nbp")
    for tag, eps, max_it in NBP_VARIANTS:
        key = method + tag
        if key + "_theta" not in g.files:
            continue
        gth, git = g[key + "_theta"], g[key + "_iterations"]
        for libm in ("glibc", "device"):                     # (on these fits the device library's erf / exp end on the same float32 values)
            be.set_mle_mode("strict")
            be.set_mle_libm(libm)
            try:
                th, cr, ll, it = be.gaussmle_arrays(d["spots"], eps, max_it, method)
            finally:
                be.set_mle_mode("refit")
                be.set_mle_libm("auto")
            assert np.array_equal(it, git), (name, key, libm)
            same = np.array([np.array_equal(a, b, equal_nan=True) for a, b in zip(th, gth)])
            assert same.all(), (name, key, libm, np.flatnonzero(~same)[:8])
        th, cr, ll, it = be.gaussmle_arrays(d["spots"], eps, max_it, method)
        if name != "degenerate7":
            _check_fit(th, cr, ll, it, gth, g[key + "_crlb"], g[key + "_loglik"], git, (), max_it)
            continue
        # degenerate7: the same rows, no row loose — but the CRLB of a (near-)singular Fisher matrix is what the SVD leaves
        # of the cut-off singular values (LAPACK in the goldens: 1.3e-29 or -4.5e-7 where the exact entry is 0), so it is
        # bounded by 1e-5 of the row's largest entry as in tests/test_oracle_golden.py, not relatively
        assert_mle_rows(th[:, 0], th[:, 1], th[:, 4], th[:, 5], th[:, 2], it, gth[:, 0], gth[:, 1], gth[:, 4], gth[:, 5],
                        gth[:, 2], git, max_it, label=f"{name} {key}")
        ref = g[key + "_crlb"]
        conv = git < max_it
        with np.errstate(invalid="ignore"):
            diff = np.abs(cr.astype(np.float64) - ref)
        diff[(cr == ref) | (np.isnan(cr) & np.isnan(ref))] = 0.0
        tol = 4e-3 * np.abs(ref) + 1e-5 * np.nanmax(np.abs(ref), axis=1, keepdims=True)
        assert np.all(diff[conv] <= tol[conv]), (name, key)


@pytest.mark.parametrize("box", [5, 7, 9, 11, 13, 15, 17, 19, 21])
def test_gaussmle_all_boxes_vs_oracle(be, orc, box):
    from math import erf, sqrt
    rng = np.random.default_rng(box)
    n, c = 96, box // 2
    spots = np.empty((n, box, box), np.float32)
    idx = np.arange(box)
    for i in range(n):
        x0, y0 = c + rng.uniform(-0.8, 0.8), c + rng.uniform(-0.8, 0.8)
        sx, sy = rng.uniform(0.9, 0.25 * box), rng.uniform(0.9, 0.25 * box)
        ex = np.array([0.5 * (erf((k - x0 + .5) / (sqrt(2) * sx)) - erf((k - x0 - .5) / (sqrt(2) * sx))) for k in idx])
        ey = np.array([0.5 * (erf((k - y0 + .5) / (sqrt(2) * sy)) - erf((k - y0 - .5) / (sqrt(2) * sy))) for k in idx])
        spots[i] = rng.poisson(rng.uniform(1500, 9000) * np.outer(ey, ex) + rng.uniform(2, 30))
    for method in ("sigmaxy", "sigma"):
        a = be.gaussmle_arrays(spots, 1e-3, 100, method)
        b = orc.gaussmle(spots, 1e-3, 100, method, threads=4)
        _check_fit(*a, *b)


def _adversarial_spots(box, n, seed):
    """20...9000 photons on backgrounds 0.5...30, centres up to 1.5 px off, sigma 0.6 px...0.25 box + 0.3: a
    fifth of these fits run into max_it, many leave the box or collapse to the sigma floor."""
    from math import erf, sqrt
    rng = np.random.default_rng(seed)
    c, idx = box // 2, np.arange(box)
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        x0, y0 = c + rng.uniform(-1.5, 1.5), c + rng.uniform(-1.5, 1.5)
        sx, sy = rng.uniform(0.6, 0.25 * box + 0.3), rng.uniform(0.6, 0.25 * box + 0.3)
        ex = np.array([0.5 * (erf((k - x0 + .5) / (sqrt(2) * sx)) - erf((k - x0 - .5) / (sqrt(2) * sx))) for k in idx])
        ey = np.array([0.5 * (erf((k - y0 + .5) / (sqrt(2) * sy)) - erf((k - y0 - .5) / (sqrt(2) * sy))) for k in idx])
        spots[i] = rng.poisson(rng.uniform(20, 9000) * np.outer(ey, ex) + rng.uniform(0.5, 30))
    return spots


@pytest.mark.parametrize("box", [3, 5, 7, 9, 13, 15, 17, 21])
@pytest.mark.parametrize("method", ["sigmaxy", "sigma"])
def test_gaussmle_strict_mode_is_the_oracle_bit_for_bit(be, orc, box, method):
    """PMI_MLE_STRICT runs the Newton loop in the reference's arithmetic (float64 intermediates, float32 stores,
    sequential accumulation): theta and the iteration count of EVERY spot — diverging, floored and max_it ones
    included — equal the oracle's bit for bit."""
    spots = _adversarial_spots(box, 1200, 1000 + box)
    o = orc.gaussmle(spots, 1e-3, 100, method, threads=4)
    be.set_mle_mode("strict")
    try:
        g = be.gaussmle_arrays(spots, 1e-3, 100, method)
    finally:
        be.set_mle_mode("refit")
    assert np.array_equal(g[3], o[3])
    assert np.array_equal(g[0].view(np.uint32), o[0].view(np.uint32))


@pytest.mark.parametrize("n", [1, 5, 67, 1031])
@pytest.mark.parametrize("max_it,eps", [(0, 1e-3), (1, 1e-3), (3, 1e-5), (100, 1e-1), (250, 1e-6)])
def test_gaussmle_strict_mode_at_the_knobs(be, orc, n, max_it, eps):
    """The all-strict launch (start-value kernel, then lane groups that take their next spot when their fit ends,
    csrc/gaussmle_strict.hip) at the edges of its bookkeeping: batches smaller than a wavefront's four groups and not a
    multiple of them, max_it 0 (the start values are the result: no group ever iterates), fits that stop after one step and
    fits that run long — theta and iteration count of every spot are the oracle's bit for bit, both methods, two boxes."""
    for box, method in ((7, "sigmaxy"), (13, "sigma")):
        spots = _adversarial_spots(box, n, 4000 + 7 * n + max_it)
        o = orc.gaussmle(spots, eps, max_it, method, threads=4)
        be.set_mle_mode("strict")
        try:
            g = be.gaussmle_arrays(spots, eps, max_it, method)
        finally:
            be.set_mle_mode("refit")
        assert np.array_equal(g[3], o[3]), (box, method)
        assert np.array_equal(g[0].view(np.uint32), o[0].view(np.uint32)), (box, method)


@pytest.mark.parametrize("box,method", [(7, "sigmaxy"), (7, "sigma"), (5, "sigmaxy"), (13, "sigmaxy"), (17, "sigma")])
def test_gaussmle_default_mode_every_row_on_adversarial_spots(be, orc, box, method):
    """The default mode (float32 loop + re-fit of flagged spots) on ill-conditioned input: no row is exempt — equal
    iteration counts everywhere, tolerance wherever the reference converged; the re-fitted rows carry the oracle's
    bits; and the float32 loop alone ("fast") does NOT pass this, which is what the flags are for."""
    spots = _adversarial_spots(box, 20000, 77 + box)
    o = orc.gaussmle(spots, 1e-3, 100, method, threads=orc.max_threads())
    g = be.gaussmle_arrays(spots, 1e-3, 100, method)
    refit = be.last_refit_count()
    assert 0 < refit < len(spots)
    assert_mle_rows(g[0][:, 0], g[0][:, 1], g[0][:, 4], g[0][:, 5], g[0][:, 2], g[3],
                    o[0][:, 0], o[0][:, 1], o[0][:, 4], o[0][:, 5], o[0][:, 2], o[3], label=f"adversarial {box} {method}")
    bit = np.all(g[0].view(np.uint32) == o[0].view(np.uint32), axis=1)
    assert bit.sum() >= refit                                  # every re-fitted row is the oracle's, bit for bit
    assert np.all(bit[o[3] >= 100])                            # max_it rows are always re-fitted
    be.set_mle_mode("fast")
    try:
        f = be.gaussmle_arrays(spots, 1e-3, 100, method)
    finally:
        be.set_mle_mode("refit")
    assert (f[3] != o[3]).sum() > 0 and be.get_mle_mode()[0] == "refit"


def test_gaussmle_mode_api(be):
    assert be.get_mle_mode() == ("refit", 0.001)
    with pytest.raises(ValueError):
        be.set_mle_mode("nope")
    from picasso_amd import _lib
    assert _lib.load().pmi_mle_set_mode(7, 0.001) != 0 and "unknown MLE mode" in _lib.last_error()
    assert _lib.load().pmi_mle_set_mode(1, 1.5) != 0
    assert be.get_mle_mode() == ("refit", 0.001)


def test_gaussmle_edge_cases(be):
    th, cr, ll, it = be.gaussmle_arrays(np.zeros((0, 7, 7), np.float32), 1e-3, 100)
    assert th.shape == (0, 6) and it.shape == (0,)
    with pytest.raises(ValueError, match="Method not available"):
        be.gaussmle_arrays(np.zeros((1, 7, 7), np.float32), 1e-3, 100, "nope")
    d = golden("gaussmle_poisson5")
    th, cr, ll, it = be.gaussmle_arrays(d["spots"][:5], 1e-3, 3)
    assert it.max() <= 3


def test_localize_pipeline_on_resident_movie(be, orc, testdata_movie):
    """identify -> fused cut+fit -> table, all on device, vs oracle composition."""
    cam = {"Baseline": 100.0, "Sensitivity": 0.5, "Gain": 2.0}
    dm = be.DeviceMovie(testdata_movie)
    try:
        t = be.localize_mle_device(dm.ptr, dm.dtype, dm.shape, 7, 1500, cam)
    finally:
        dm.free()
    fr, y, x, ng = orc.identify(testdata_movie, 1500, 7)
    spots = orc.get_spots(testdata_movie, fr, y, x, 7, cam)
    th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy")
    assert len(t["frame"]) == len(fr)
    assert np.array_equal(t["frame"], fr.astype(np.uint32))
    assert np.array_equal(t["net_gradient"], ng)
    assert_mle_rows(t["x"], t["y"], t["sx"], t["sy"], t["photons"], t["iterations"],
                    th[:, 0] + x - 3, th[:, 1] + y - 3, th[:, 4], th[:, 5], th[:, 2], it)
    assert np.max(np.abs(t["photons"] - th[:, 2]) / th[:, 2]) < 1e-4


@pytest.mark.parametrize("shape,box", [((3, 96, 128), 7), ((2, 200, 520), 7), ((2, 70, 1032), 5), ((1, 130, 64), 9),
                                      ((4, 64, 16), 3), ((2, 137, 512), 7), ((2, 66, 1536), 9), ((3, 40, 24), 7),
                                      ((3, 90, 130), 7), ((2, 75, 518), 5), ((2, 64, 36), 7), ((1, 131, 1030), 9),
                                      ((2, 150, 520), 11), ((2, 137, 512), 13), ((2, 90, 130), 13), ((1, 70, 1032), 11),
                                      ((3, 64, 40), 13), ((2, 120, 520), 15), ((2, 90, 130), 17), ((1, 70, 1032), 17), ((2, 64, 48), 15)])
def test_identify_fast_path_vs_oracle(be, orc, shape, box):
    """uint16 movies of even width take the register-pipelined scan (identify_fast.hip): multi-segment
    rows, partial bands, partial last segment, widths that are not a multiple of 8 (the last chunk of a
    row runs into the next row / past the movie), ties, saturated plateaus, ROI crops, low/high thresholds."""
    rng = np.random.default_rng(hash((shape, box)) % 2**32)
    mov = rng.poisson(30, size=shape).astype(np.uint16) + 100
    F, Y, X = shape
    for f in range(F):                                   # bright blobs, some on the border bands
        for _ in range(max(2, Y * X // 1500)):
            y, x = int(rng.integers(1, Y - 1)), int(rng.integers(1, X - 1))
            amp = int(rng.integers(200, 4000))
            mov[f, max(0, y - 1):y + 2, max(0, x - 1):x + 2] += np.uint16(amp // 3)
            mov[f, y, x] += np.uint16(amp)
    mov[0, 10:14, 8:14] = 65535                          # saturated plateau (ties at the clamp value)
    mov[-1, Y // 2, : X // 2] = 65535
    mov[:, ::9, ::4] = mov[:, ::9, ::4] // 8 * 8         # ties
    for min_ng in (-1e9, 300.0, 20000.0):
        a = be.identify_arrays(mov, min_ng, box)
        b = orc.identify(mov, min_ng, box, threads=4)
        assert len(a[0]) == len(b[0]), (min_ng, len(a[0]), len(b[0]))
        assert all(np.array_equal(p, q) for p, q in zip(a, b))
    if X >= 32:
        # crops: aligned to 8 columns, and starting / ending anywhere (the fast path works on frame-aligned
        # chunks and masks what lies outside the crop; the wrapped stencil column wraps inside the CROP)
        for roi in (((3, 8), (Y - 2, X - 8)), ((0, 3), (Y, X - 5)), ((2, 1), (Y - 1, X)), ((5, 13), (Y - 3, X - 2)),
                    ((1, 7), (Y - 4, 7 + 17))):
            for min_ng in (-1e9, 300.0):
                a = be.identify_arrays(mov, min_ng, box, roi=roi)
                b = orc.identify(mov, min_ng, box, roi=roi, threads=4)
                assert len(a[0]) == len(b[0]), (roi, min_ng)
                assert all(np.array_equal(p, q) for p, q in zip(a, b)), (roi, min_ng)


@pytest.mark.parametrize("shape,box", [((5, 128, 128), 7), ((3, 200, 128), 7), ((2, 300, 100), 7), ((2, 520, 256), 7), ((3, 70, 250), 7),
                                       ((2, 33, 64), 7), ((3, 260, 120), 7), ((4, 97, 130), 5), ((2, 300, 128), 5), ((2, 150, 256), 9),
                                       ((3, 280, 96), 9), ((2, 41, 200), 9), ((6, 64, 64), 5), ((2, 200, 256), 13), ((3, 90, 130), 11),
                                       ((2, 150, 120), 13), ((2, 260, 250), 11), ((7, 64, 64), 7), ((3, 130, 60), 7), ((2, 300, 48), 7), ((4, 40, 34), 7)])
def test_identify_narrow_frames_packed_bands(be, orc, shape, box):
    """Frames at most 256 (128) pixels wide put two (four) bands of a frame side by side in one wavefront
    (boxes 5, 7, 9): heights that leave sub-bands partly or wholly empty, crops that start off the 8-pixel grid,
    maxima on the first / last allowed rows and columns of every sub-band, a saturated plateau (overflow rescan)."""
    rng = np.random.default_rng(shape[1] * 1000 + shape[2] + box)
    F, Y, X = shape
    mov = rng.poisson(25, size=shape).astype(np.uint16) + 90
    h = box // 2
    for f in range(F):
        for _ in range(max(3, Y * X // 900)):
            y, x = int(rng.integers(1, Y - 1)), int(rng.integers(1, X - 1))
            amp = int(rng.integers(150, 3000))
            mov[f, max(0, y - 1):y + 2, max(0, x - 1):x + 2] += np.uint16(amp // 3)
            mov[f, y, x] += np.uint16(amp)
        for y in (h, h + 1, 31, 32, 33, 63, 64, 65, Y - h - 2, Y - h - 3):          # band seams and frame edges
            for x in (h, X - h - 2, X // 2):
                if 0 < y < Y - 1:
                    mov[f, y, x] += np.uint16(2500 + 7 * y + x)
    mov[0, 5:9, 4:10] = 65535
    mov[:, ::11, ::3] = mov[:, ::11, ::3] // 4 * 4
    for min_ng in (-1e9, 400.0, 15000.0):
        a = be.identify_arrays(mov, min_ng, box)
        b = orc.identify(mov, min_ng, box, threads=4)
        assert len(a[0]) == len(b[0]), (min_ng, len(a[0]), len(b[0]))
        assert all(np.array_equal(p, q) for p, q in zip(a, b)), min_ng
    for roi in (((0, 3), (Y, X - 1)), ((2, 5), (Y - 3, X - 6)), ((1, 9), (Y - 1, min(X, 9 + 101))), ((7, 0), (Y - 2, X // 2 + 3))):
        a = be.identify_arrays(mov, 400.0, box, roi=roi)
        b = orc.identify(mov, 400.0, box, roi=roi, threads=4)
        assert len(a[0]) == len(b[0]), roi
        assert all(np.array_equal(p, q) for p, q in zip(a, b)), roi
    sat = np.full((2, Y, X), 65535, np.uint16)                  # every pixel a packed-test candidate: the rescan path
    sat[1, Y // 2, X // 2] = 100
    a = be.identify_arrays(sat, -1e9, box)
    b = orc.identify(sat, -1e9, box, threads=4)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.int16])
@pytest.mark.parametrize("shape,box", [((19, 64, 64), 7), ((9, 40, 34), 7), ((32, 64, 64), 7), ((13, 128, 128), 7), ((11, 97, 130), 5),
                                       ((6, 150, 256), 9), ((21, 64, 64), 5), ((17, 33, 64), 7), ((10, 30, 200), 13), ((4, 64, 64), 7)])
def test_identify_narrow_short_frames_side_by_side(be, orc, shape, box, dtype):
    """Round 5: a narrow frame that is also SHORT fills the wavefront with the same rows of 2 / 4 / 8 consecutive frames
    (identify_fast.hip, FastParams::pf) instead of row ranges of one frame, each of which pays its halo rows — taken when
    the cost model says so (64 x 64 at box 7: 2.3 -> 4.1 TB/s).  Frame counts that are not a multiple of the lane sets
    (the last group repeats its last frame and marks nothing), fewer frames than lane sets (the other form), maxima on
    the first / last allowed rows and columns, ties, a saturated frame in the middle of a group (the rescan path walks
    the group's frames), crops off the 8-pixel grid, frame bounds that start inside a group; bit-exact against the oracle
    (picasso/localize.py:97-134, 202-244, 288).  These short movies exercise the dispatch around the new form;
    test_identify_frames_side_by_side_on_long_movies runs the form itself."""
    rng = np.random.default_rng(shape[0] * 7919 + shape[1] * 31 + shape[2] + box)
    F, Y, X = shape
    h = box // 2
    top = 250 if dtype == np.uint8 else (30000 if dtype == np.int16 else 60000)
    mov = (rng.poisson(12, size=shape) + (3 if dtype == np.uint8 else 90)).astype(np.int64)
    for f in range(F):
        for _ in range(max(3, Y * X // 700)):
            y, x = int(rng.integers(1, Y - 1)), int(rng.integers(1, X - 1))
            amp = int(rng.integers(30, 200)) if dtype == np.uint8 else int(rng.integers(150, 3000))
            mov[f, max(0, y - 1):y + 2, max(0, x - 1):x + 2] += amp // 3
            mov[f, y, x] += amp
        for y in (h, h + 1, Y - h - 2, Y - h - 3):
            for x in (h, X - h - 2, X // 2):
                mov[f, y, x] += (60 if dtype == np.uint8 else 2500) + (7 * y + x + 3 * f) % 40
    mov[:, ::11, ::3] = mov[:, ::11, ::3] // 4 * 4
    if dtype == np.int16:
        mov -= 400                                       # negative counts (a baseline set too high)
    mov = np.clip(mov, np.iinfo(dtype).min, top).astype(dtype)
    if F > 5:
        mov[F // 2] = top                                # a saturated frame inside a group of frames
        mov[F // 2, Y // 2, X // 2] = 1
    lo_t = 40.0 if dtype == np.uint8 else 400.0
    for min_ng in (-1e9, lo_t, 30 * lo_t):
        a = be.identify_arrays(mov, min_ng, box)
        b = orc.identify(mov, min_ng, box, threads=4)
        assert len(a[0]) == len(b[0]) and (min_ng > 0 or len(b[0]) > 20), (min_ng, len(a[0]), len(b[0]))
        assert all(np.array_equal(p, q) for p, q in zip(a, b)), min_ng
    # which form ran: the library names the kernel instance it launched last (" frames" = frames side by side)
    import ctypes
    from picasso_amd import _lib
    name = ctypes.create_string_buffer(128)
    _lib.load().pmi_last_scan_kernel(name, 128)
    # (a few frames: the cost model keeps the row ranges of one frame — its measure is the busiest wavefront, and there are
    # more wavefronts than frames; the long movies below take the other form)
    assert b" frames" not in name.value, name.value
    if X >= 32:
        for roi in (((0, 3), (Y, X - 1)), ((2, 5), (Y - 3, X - 6)), ((1, 9), (Y - 1, min(X, 9 + 40)))):
            a = be.identify_arrays(mov, lo_t, box, roi=roi)
            b = orc.identify(mov, lo_t, box, roi=roi, threads=4)
            assert all(np.array_equal(p, q) for p, q in zip(a, b)), roi
    for fb in ((3, F - 2), (1, 1), (F - 1, F - 1)):
        if fb[0] <= fb[1] < F:
            a = be.identify_arrays(mov, lo_t, box, frame_bounds=fb)
            b = orc.identify(mov, lo_t, box, frame_bounds=fb, threads=4)
            assert all(np.array_equal(p, q) for p, q in zip(a, b)), fb


@pytest.mark.parametrize("shape,box,dtype", [((40003, 64, 64), 7, np.uint16), ((33001, 40, 50), 7, np.uint8), ((41003, 36, 100), 7, np.int16),
                                             ((40001, 30, 120), 5, np.uint16), ((24001, 40, 250), 9, np.uint16)])
def test_identify_frames_side_by_side_on_long_movies(be, orc, shape, box, dtype):
    """The frames-side-by-side form of the packed scan on movies long enough for the cost model to take it (more groups
    of frames than persistent wavefronts): 8 / 4 / 2 lane sets, a last group that is not full, every pixel type of the
    packed scan, a saturated frame inside a group, frame bounds and a crop; the kernel instance that ran is asserted, the
    identifications are the oracle's bit for bit."""
    import ctypes
    from picasso_amd import _lib
    rng = np.random.default_rng(shape[0] + box)
    F, Y, X = shape
    top = 250 if dtype == np.uint8 else (30000 if dtype == np.int16 else 60000)
    mov = (rng.poisson(10, size=shape) + (3 if dtype == np.uint8 else 90)).astype(np.int32)
    n_spots = F * max(1, Y * X // 2000)
    ff, yy, xx = rng.integers(0, F, n_spots), rng.integers(1, Y - 1, n_spots), rng.integers(1, X - 1, n_spots)
    amp = rng.integers(40, 200, n_spots) if dtype == np.uint8 else rng.integers(300, 4000, n_spots)
    np.add.at(mov, (ff, yy, xx), amp)
    np.add.at(mov, (ff, yy - 1, xx), amp // 3)
    np.add.at(mov, (ff, yy, xx + 1), amp // 3)
    if dtype == np.int16:
        mov -= 300
    mov = np.clip(mov, np.iinfo(dtype).min, top).astype(dtype)
    mov[F // 3] = top
    mov[F // 3, Y // 2, X // 2] = 1
    lo_t = 60.0 if dtype == np.uint8 else 800.0
    name = ctypes.create_string_buffer(128)
    for kw in ({}, {"frame_bounds": (5, F - 4)}, {"roi": ((1, 3), (Y - 2, X - 5))}):
        for min_ng in ((lo_t, 25 * lo_t) if not kw else (lo_t,)):
            a = be.identify_arrays(mov, min_ng, box, **kw)
            _lib.load().pmi_last_scan_kernel(name, 128)
            assert name.value.decode().endswith(" frames"), (kw, name.value)
            b = orc.identify(mov, min_ng, box, threads=orc.max_threads(), **kw)
            assert len(a[0]) == len(b[0]) and (min_ng > lo_t or len(b[0]) > F // 4), (kw, min_ng, len(a[0]), len(b[0]))
            assert all(np.array_equal(p, q) for p, q in zip(a, b)), (kw, min_ng)


def test_identify_fast_path_all_zero_and_all_saturated(be, orc):
    for val in (0, 65535, 777):
        mov = np.full((2, 80, 256), val, np.uint16)
        a = be.identify_arrays(mov, -1e9, 7)
        b = orc.identify(mov, -1e9, 7, threads=2)
        assert len(a[0]) == len(b[0]) == 0
    mov = np.zeros((1, 80, 256), np.uint16)
    mov[0, 40, 100] = 65535; mov[0, 40, 101] = 65535; mov[0, 3, 3] = 9; mov[0, 75, 251] = 9
    a = be.identify_arrays(mov, -1e9, 7)
    b = orc.identify(mov, -1e9, 7, threads=2)
    assert all(np.array_equal(p, q) for p, q in zip(a, b)) and len(a[0]) >= 3


@pytest.mark.parametrize("dtype,box", [(np.uint16, 7), (np.uint16, 9), (np.uint16, 13), (np.uint8, 7), (np.int16, 5)])
def test_identify_plateaus_and_saturated_fiducials(be, orc, dtype, box):
    """Plateaus of equal pixels (a saturated fiducial, flat regions of every shape, equal pixels side by side and one
    above the other across lane and flush-group boundaries): np.argmax takes the FIRST maximum of a window, the scan drops
    right-hand / lower neighbours of a candidate before the exact test, the result stays the reference's."""
    rng = np.random.default_rng(box * 7 + np.dtype(dtype).itemsize)
    top = {np.uint16: 65535, np.uint8: 255, np.int16: 32767}[dtype]
    base = rng.poisson(30, size=(6, 200, 512)).astype(np.int64) + (0 if dtype != np.int16 else -200)
    mov = np.clip(base, np.iinfo(dtype).min, top).astype(dtype)
    mov[:, 40:90, 100:170] = top                               # saturated square, lane-unaligned
    mov[:, 120:123, 5:507] = top // 2                          # a long flat bar over every lane
    mov[0:3, 150:190, 300:303] = top // 3                      # a tall narrow one
    mov[3:, 20:28, 7:9] = top                                  # pairs side by side at a lane boundary (columns 7 | 8)
    mov[:, 99:101, 400] = top - 1                              # one above the other
    for f in range(6):                                         # plateaus with a brighter pixel inside / at the rim
        mov[f, 60, 130 + f] = top if dtype != np.uint16 else top
        mov[f, 160 + f, 301] = top // 3 + 5
    for min_ng in (-1e9, 200.0, 0.3 * float(top)):
        a = be.identify_arrays(mov, min_ng, box)
        b = orc.identify(mov, min_ng, box, threads=4)
        assert all(np.array_equal(p, q) for p, q in zip(a, b)), (dtype, box, min_ng, len(a[0]), len(b[0]))
    assert len(b[0]) > 0


@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.uint32])
def test_identify_wide_movies_holding_counts(be, orc, dtype):
    """32-bit movies whose pixels are 16-bit counts, clean and with pixels that are not such counts (a fraction, a negative,
    65536, NaN).  Rounds 1 - 5 narrowed the integer ones to uint16 chunk by chunk (a spoiled chunk took the generic kernel);
    since round 6 they take the key scan like float32 movies, and the chunked route serves only frames too narrow for it —
    the table is the reference's either way, chunk boundaries included."""
    from picasso_amd import _lib
    assert _lib.load().pmi_identify_set_narrow_chunk(3) == 0          # several chunks on a small movie
    try:
        _wide_movie_cases(be, orc, dtype)
    finally:
        assert _lib.load().pmi_identify_set_narrow_chunk(0) == 0


def _wide_movie_cases(be, orc, dtype):
    rng = np.random.default_rng(21)
    base = rng.poisson(40, size=(11, 96, 272)).astype(np.float64)
    for f in range(11):
        for _ in range(6):
            y, x = rng.integers(8, 88), rng.integers(8, 264)
            yy, xx = np.mgrid[y - 4:y + 5, x - 4:x + 5]
            base[f, y - 4:y + 5, x - 4:x + 5] += np.rint(3000 * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / 1.3 ** 2))
    clean = base.astype(dtype)
    spoiled = clean.copy()
    if dtype == np.float32:
        spoiled[4, 50, 100] += 0.5            # chunk 1 (frames 3..5): a fraction
        spoiled[7, 3, 3] = np.float32(-1.0)   # chunk 2: negative
        spoiled[10, 90, 200] = 70000.0        # chunk 3: beyond 16 bits
    elif dtype == np.int32:
        spoiled[4, 50, 100] = -7
        spoiled[10, 90, 200] = 65536
    else:
        spoiled[4, 50, 100] = 65536
        spoiled[9, 1, 1] = 4000000000
    for mov in (clean, spoiled):
        for min_ng, roi in ((5000.0, None), (-1e9, None), (3000.0, ((5, 9), (90, 250)))):
            a = be.identify_arrays(mov, min_ng, 7, roi=roi)
            b = orc.identify(mov, min_ng, 7, roi=roi, threads=4)
            assert all(np.array_equal(p, q) for p, q in zip(a, b)), (dtype, min_ng, roi, len(a[0]), len(b[0]))
    assert len(b[0]) > 20


@pytest.mark.parametrize("kind", ["fractions", "negative_offset", "tiny_scale", "nan_inf", "ties", "mixed_chunks"])
@pytest.mark.parametrize("box", [5, 7, 9, 11, 13, 15, 17])
def test_identify_float32_movies_with_any_content(be, orc, kind, box):
    """float32 movies that are not 16-bit counts (the reference treats every movie as float32, picasso/localize.py:332): the
    packed scan runs on 16-bit keys — the upper half of the order-preserving integer image of a float32 — and the first-argmax
    rule, the float32 net gradient and the threshold are decided on the float32 pixels (identify_fast.hip PT_KEY).  Fractions,
    a negative offset, values far below 1, NaN / +-inf pixels, plateaus of equal values and chunks of either kind in one
    movie: the table is the oracle's, bit for bit."""
    from picasso_amd import _lib
    rng = np.random.default_rng(300 + box)
    F, Y, X = 7, 120, 336
    mov = rng.poisson(35, size=(F, Y, X)).astype(np.float64)
    for f in range(F):
        for _ in range(10):
            y, x = rng.integers(9, Y - 9), rng.integers(9, X - 9)
            s = rng.uniform(0.9, 1.0 + 0.15 * box)
            yy, xx = np.mgrid[y - 8:y + 9, x - 8:x + 9]
            mov[f, y - 8:y + 9, x - 8:x + 9] += rng.uniform(800, 5000) * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / s ** 2)
    min_ngs = [3000.0, 300.0, -1e9]
    if kind == "fractions":
        mov = mov * 1.37 + 0.25
    elif kind == "negative_offset":
        mov = mov * 0.731 - 5000.0
    elif kind == "tiny_scale":
        mov = mov * 1e-3
        min_ngs = [3.0, 0.3, -1e9]
    elif kind == "nan_inf":
        mov = mov + 0.5
        for v in (np.nan, np.inf, -np.inf, -np.nan):
            for _ in range(12):
                mov[rng.integers(0, F), rng.integers(0, Y), rng.integers(0, X)] = v
    elif kind == "ties":
        mov = np.round(mov / 16.0) * 16.0 + 0.5             # many equal neighbours, equal keys and equal pixels
        mov[2, 40:60, 100:140] = 7000.5                     # a plateau
    mov = mov.astype(np.float32)
    if kind == "mixed_chunks":
        mov[:3] = np.rint(mov[:3])                          # frames 0..2 hold counts (narrowed exactly), the others fractions
        mov[3:] += 0.125
    assert _lib.load().pmi_identify_set_narrow_chunk(3) == 0          # several chunks on a small movie
    try:
        n_last = 0
        for min_ng in min_ngs:
            for roi in (None, ((5, 11), (Y - 3, X - 13))):
                a = be.identify_arrays(mov, min_ng, box, roi=roi)
                b = orc.identify(mov, min_ng, box, roi=roi, threads=4)
                assert len(a[0]) == len(b[0]) and all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b)), \
                    (kind, box, min_ng, roi, len(a[0]), len(b[0]))
                n_last = len(b[0])
        assert n_last > 20
    finally:
        assert _lib.load().pmi_identify_set_narrow_chunk(0) == 0


@pytest.mark.parametrize("dtype", [np.int32, np.uint32])
@pytest.mark.parametrize("box", [5, 7, 9, 11, 13, 15, 17])
def test_identify_32bit_integer_movies_through_the_key_scan(be, orc, dtype, box):
    """32-bit integer movies compare as float32 in the reference (the frame is cast first, picasso/localize.py:332).  Round 6:
    they take the packed scan on 16-bit keys like float32 movies, with that cast in front of the keys and of every exact
    read (identify_fast.hip PT_KEY_I32 / PT_KEY_U32) — counts, values far beyond 16 bits, values beyond 2^24 (where the cast
    rounds and makes ties the integers did not have), negative pixels, and the top of the uint32 range: the oracle's
    table bit for bit, with and without an ROI, frames wider than one wavefront's 512 columns included."""
    from picasso_amd import _lib
    rng = np.random.default_rng(900 + box + (7 if dtype == np.uint32 else 0))
    for (F, Y, X) in ((5, 96, 272), (2, 70, 1100)):
        base = rng.poisson(60, size=(F, Y, X)).astype(np.int64)
        for f in range(F):
            for _ in range(max(6, X // 40)):
                y, x = rng.integers(10, Y - 10), rng.integers(10, X - 10)
                s = rng.uniform(0.9, 1.0 + 0.12 * box)
                yy, xx = np.mgrid[y - 9:y + 10, x - 9:x + 10]
                base[f, y - 9:y + 10, x - 9:x + 10] += np.rint(rng.uniform(800, 5000) * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / s ** 2)).astype(np.int64)
        kinds = {"counts": (base, 3000.0), "x30011": (base * 30011, 3000.0 * 30011), "beyond_2^24": (base * 4096 + 16777000, 3000.0 * 4096)}
        if dtype == np.int32:
            kinds["negative"] = (base * 1000 - 2_000_000_000, 3000.0 * 1000)
        else:
            kinds["top_of_range"] = (base * 1000 + 4_200_000_000, 3000.0 * 1000)
        for kind, (m64, min_ng) in kinds.items():
            assert m64.min() >= np.iinfo(dtype).min and m64.max() <= np.iinfo(dtype).max
            mov = m64.astype(dtype)
            for roi in (None, ((3, 9), (Y - 2, X - 11))):
                for t in (min_ng, -1e18):
                    a = be.identify_arrays(mov, t, box, roi=roi)
                    b = orc.identify(mov, t, box, roi=roi, threads=4)
                    assert len(a[0]) == len(b[0]) and all(np.array_equal(p, q) for p, q in zip(a, b)), (kind, dtype, box, roi, t, len(a[0]), len(b[0]))
            if X >= 512:        # (frames of at most 256 columns pack two row ranges per wavefront and keep the chunked route)
                import ctypes
                name = ctypes.create_string_buffer(128)
                _lib.load().pmi_last_scan_kernel(name, 128)
                assert b"identify_scan_u16_fast_kernel" in name.value and (b", 4, " in name.value or b", 5, " in name.value), name.value
    assert len(b[0]) > 10


@pytest.mark.parametrize("box", [9, 13, 15, 17])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_identify_tie_dense_low_count_movies(be, orc, dtype, box):
    """A quantised low-count background (Poisson(15): half a dozen distinct values) at a threshold the floor cannot bite on: one
    or two pixels in a hundred tie for the maximum of their window and fill the wave's candidate ring within a few rows.  Round 6:
    the ring is looked at after every flush group (a chunk ends with the group that leaves fewer than 128 free entries) — before,
    at boxes 15 / 17 an unroll period's rows overflowed it and the chunk was rescanned pixel by pixel (the path of saturated
    fiducials: the same table, at 20 GB/s).  The table is the oracle's bit for bit either way; this keeps the ring-full path
    under test: frames wider than a wavefront's 512 columns too."""
    rng = np.random.default_rng(77 + box)
    for (F, Y, X) in ((4, 150, 512), (2, 90, 1100)):
        mov = rng.poisson(15, size=(F, Y, X)).astype(np.float64)
        for f in range(F):
            for _ in range(8):
                y, x = rng.integers(12, Y - 12), rng.integers(12, X - 12)
                yy, xx = np.mgrid[y - 9:y + 10, x - 9:x + 10]
                mov[f, y - 9:y + 10, x - 9:x + 10] += np.rint(rng.uniform(60, 200) * np.exp(-0.5 * ((yy - y) ** 2 + (xx - x) ** 2) / 1.4 ** 2))
        mov = np.minimum(mov, 255).astype(dtype)
        for min_ng in (600.0, 150.0, -1e9):
            a = be.identify_arrays(mov, min_ng, box)
            b = orc.identify(mov, min_ng, box, threads=4)
            assert len(a[0]) == len(b[0]) and all(np.array_equal(p, q) for p, q in zip(a, b)), (dtype, box, (F, Y, X), min_ng, len(a[0]), len(b[0]))
    assert len(b[0]) > 100


def test_identify_capacity_retry(be, orc, testdata_movie):
    """More rows than the first capacity guess: PMI_ERR_CAPACITY -> retry with the exact count."""
    rng = np.random.default_rng(5)
    mov = rng.integers(0, 4000, size=(3, 256, 256)).astype(np.uint16)
    a = be.identify_arrays(mov, -1e9, 3)       # ~1/9 of all pixels are maxima: far above 4096 rows
    b = orc.identify(mov, -1e9, 3, threads=4)
    assert len(a[0]) == len(b[0]) > 4096
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


# ---------------------------------------------------------------------------
# gausslq: MINPACK lmdif per spot (picasso/gausslq.py:206-300)
# ---------------------------------------------------------------------------
def _check_lq(th, info, nfev, oth, oinfo, onfev, exact_frac=0.9995):
    """The kernel follows the oracle's float64 lmdif operation by operation except for the order of the sums over
    the box, and every spot on which that order can matter — a decision of lmdif taken within rounding distance of
    its threshold — is fitted again with MINPACK's order: theta (float32), info and nfev coincide to the last bit
    on (practically) every spot, and all spots stay inside the north-star tolerance."""
    assert th.shape == oth.shape
    exact = np.all((th == oth) | (np.isnan(th) & np.isnan(oth)), axis=1)
    assert exact.mean() >= exact_frac, f"only {exact.mean():.4f} bit-identical"
    assert (info == oinfo).mean() >= exact_frac and (nfev == onfev).mean() >= exact_frac
    fin = np.all(np.isfinite(oth), axis=1)
    assert np.max(np.abs(th[fin][:, [0, 1, 4, 5]] - oth[fin][:, [0, 1, 4, 5]]), initial=0) < 1e-3
    rel = np.abs(th[fin][:, 2:4] - oth[fin][:, 2:4]) / np.maximum(np.abs(oth[fin][:, 2:4]), 1.0)
    assert np.max(rel, initial=0) < 1e-3


@pytest.mark.parametrize("name", ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "poisson13"])
def test_gausslq_vs_oracle_and_goldens(be, orc, name):
    g = golden("gausslq_" + name)
    th, info, nfev = be.gausslq_arrays(g["spots"], full_output=True)
    oth, oinfo, onfev = orc.gausslq(g["spots"], full=True, threads=4)
    _check_lq(th, info, nfev, oth, oinfo, onfev)
    # the reference's own output (float32 start values, scipy MINPACK): what ftol = xtol = 1e-2 leaves open
    assert np.max(np.abs(th[:, :2] - g["theta"][:, :2])) < 5e-3
    assert np.max(np.abs(th[:, 4:] - g["theta"][:, 4:])) < 5e-3
    assert np.max(np.abs(th[:, 2] - g["theta"][:, 2]) / g["theta"][:, 2]) < 5e-3


@pytest.mark.parametrize("name", ["conftest_clean", "conftest_noisy", "testdata_real", "poisson7", "poisson13"])
def test_gausslq_vs_numba_promotion_goldens(be, name):
    """The device (strict, the default) against the reference's own gausslq.fit_spot under numba's typing — start values
    from float64 moment sums, scipy's MINPACK over the numba-typed residual function (tests/golden/*_nbp.npz): the
    float32 theta the reference stores (gausslq.py:275) is EQUAL on every row, no oracle in between."""
    s, g = golden("gausslq_" + name), golden("gausslq_" + name + "_nbp")
    assert be.get_lq_mode() == "strict"
    th = be.gausslq_arrays(s["spots"])
    assert np.array_equal(th, g["theta"].astype(np.float32))


@pytest.mark.parametrize("box", [3, 5, 7, 9, 11, 13, 15, 17, 19, 21])
def test_gausslq_all_boxes_vs_oracle(be, orc, box):
    rng = np.random.default_rng(100 + box)
    n, c = 127, box // 2          # odd: the last wavefront has an unpaired group
    idx = np.arange(box) - c
    spots = np.empty((n, box, box), np.float32)
    for i in range(n):
        x0, y0 = rng.uniform(-0.8, 0.8, 2)
        sx, sy = rng.uniform(0.7, max(0.8, 0.22 * box), 2)
        gx = np.exp(-0.5 * ((idx - x0) / sx) ** 2) / (np.sqrt(2 * np.pi) * sx)
        gy = np.exp(-0.5 * ((idx - y0) / sy) ** 2) / (np.sqrt(2 * np.pi) * sy)
        spots[i] = rng.poisson(rng.uniform(800, 9000) * np.outer(gy, gx) + rng.uniform(2, 40))
    th, info, nfev = be.gausslq_arrays(spots, full_output=True)
    oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=4)
    _check_lq(th, info, nfev, oth, oinfo, onfev)


def _lq_adversarial_spots(box, n, seed):
    """The spots of tools/fuzz_parity.py's least-squares branch: widths 0.6 px ... a quarter of the box, centres 1.2 px
    off, 100 ... 9000 photons."""
    rng = np.random.default_rng(seed)
    idx = np.arange(box) - box // 2
    x0 = rng.uniform(-1.2, 1.2, n); y0 = rng.uniform(-1.2, 1.2, n)
    sx = rng.uniform(0.6, 0.25 * box + 0.5, n); sy = rng.uniform(0.6, 0.25 * box + 0.5, n)
    gx = np.exp(-0.5 * ((idx[None] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
    gy = np.exp(-0.5 * ((idx[None] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
    return rng.poisson(rng.uniform(100, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :]
                       + rng.uniform(0.5, 60, n)[:, None, None]).astype(np.float32)


def _lq_assert_identical(th, info, nfev, oth, oinfo, onfev):
    same = (th == oth) | (np.isnan(th) & np.isnan(oth))
    assert same.all(), f"{int((~same.all(axis=1)).sum())} of {len(th)} spots differ from MINPACK's theta"
    assert np.array_equal(info, oinfo) and np.array_equal(nfev, onfev)


@pytest.mark.parametrize("box", [3, 5, 7, 9, 11, 13, 15, 17, 19, 21])
def test_gausslq_strict_mode_is_minpack_bit_for_bit(be, orc, box):
    """pmi_gausslq_set_mode(strict), the default: every sum over the residual rows (enorm, qrfac's Householder products,
    Q^T fvec) in MINPACK's sequential order -> theta, info and nfev of EVERY spot are the oracle's lmdif's to the last bit
    (the oracle is pinned against scipy.optimize.leastsq itself, tests/test_oracle_golden.py; reference call site
    picasso/gausslq.py:240-242).  Round 3's refit mode left 5e-6 of such spots up to 2.4 px away with identical `info`."""
    assert be.get_lq_mode() == "strict"
    n = 20000 if box <= 13 else 6000
    spots = _lq_adversarial_spots(box, n, 1900 + box)
    th, info, nfev = be.gausslq_arrays(spots, full_output=True)
    # (fitted twice in this mode: only a spot with a float32 rounding of its model within a few float64 ulps of a tie, which
    # the last bit of one exp would decide — those run again with exp rounded correctly; a handful per ten million)
    assert be.last_lq_refit_count() <= max(2, n // 1000)
    oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=orc.max_threads())
    _lq_assert_identical(th, info, nfev, oth, oinfo, onfev)


def test_gausslq_strict_mode_on_the_fuzz_residuals_of_the_refit_mode(be, orc):
    """The 284 spots on which the refit mode (tree sums + a second fit of the flagged) ended away from MINPACK in a
    randomised run of 15 million (tools/fuzz_parity.py 400 4243 lq with PMI_LQ_MODE=refit, round 4: 78 of them beyond
    1e-3 px, the worst 8.3 px; kept under tests/golden/lq_fuzz_regressions/): the default mode reproduces lmdif on each."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lq_fuzz_regressions", "refit_mode_residuals_box*.npz")))
    assert len(files) == 8
    total = 0
    for f in files:
        spots = np.load(f)["spots"]
        th, info, nfev = be.gausslq_arrays(spots, full_output=True)
        oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=4)
        _lq_assert_identical(th, info, nfev, oth, oinfo, onfev)
        total += len(spots)
    assert total == 284


def test_gausslq_strict_mode_where_one_exp_decides_a_float32_rounding(be, orc):
    """With every sum in MINPACK's order, 2 spots of 29.2 million still differed from the oracle (tools/fuzz_parity.py 800 91
    lq, 13 minutes; 7e-5 and 8e-5 px, `info` and `nfev` equal): one float64 exp of the Gaussian profiles differs in its last
    bit between the device's libm and glibc — both within an ulp, neither the correctly rounded function — and flips ONE
    float32 rounding of the stored model (gausslq.py:203).  docs/history/tools/probe_lq_exp.py: the oracle gives the same theta with
    libm's exp and with a correctly rounded one on both spots, i.e. the odd bit was the device's.  The strict mode now flags
    a profile value within 8 float64 ulps of a float32 rounding boundary and fits such a spot again with exp rounded
    correctly (csrc/exp_cr.h): both spots are lmdif's bit for bit."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lq_fuzz_regressions", "strict_exp_residual_box*.npz")))
    assert len(files) == 2
    for f in files:
        spots = np.load(f)["spots"]
        th, info, nfev = be.gausslq_arrays(spots, full_output=True)
        oth, oinfo, onfev = orc.gausslq(spots, full=True)
        assert be.last_lq_refit_count() == len(spots)
        _lq_assert_identical(th, info, nfev, oth, oinfo, onfev)


@pytest.mark.parametrize("box", [3, 5, 7, 9, 13, 17, 21])
def test_gausslq_strict_mode_on_extreme_pixels_among_ordinary_spots(be, orc, box):
    """Every third spot is one lmdif was not written for — pixels scaled to 1e-38 ... 3e33, a NaN or an infinity, flat,
    empty, negative, one bright pixel / row / column, pure noise — and sits in a lane group with ordinary neighbours: the
    range guards of the column-per-lane Jacobian (csrc/lq_jacobian_w.inc: columns outside (2^-64, 2^54) take the scaled
    norm, non-finite or degenerate ones the second pass) leave theta, info and nfev of ALL spots the oracle's."""
    n = 1536
    spots = _lq_adversarial_spots(box, n, 77 + box)
    rng = np.random.default_rng(box)
    c = box // 2
    for i in range(0, n, 3):
        k = (i // 3) % 16
        if k == 0: spots[i] *= np.float32(1e-30)
        elif k == 1: spots[i] *= np.float32(1e25)
        elif k == 2: spots[i, rng.integers(box), rng.integers(box)] = np.nan
        elif k == 3: spots[i, rng.integers(box), rng.integers(box)] = np.inf
        elif k == 4: spots[i] = 0
        elif k == 5: spots[i] = 7.0
        elif k == 6: spots[i] = 0; spots[i, c, c] = 1000.0
        elif k == 7: spots[i] = -spots[i]
        elif k == 8: spots[i] *= np.float32(1e-38)
        elif k == 9: spots[i] *= np.float32(3e33)
        elif k == 10: spots[i, 0, 0] = 1e30
        elif k == 11: spots[i] = 0; spots[i, 0, :] = 500.0
        elif k == 12: spots[i] = 0; spots[i, :, -1] = 500.0
        elif k == 13: spots[i] = rng.normal(0, 1, (box, box)).astype(np.float32)
        elif k == 14: spots[i, rng.integers(box), rng.integers(box)] = -np.inf
        elif k == 15: spots[i] *= np.float32(1e-20)
    with np.errstate(all="ignore"):
        th, info, nfev = be.gausslq_arrays(spots, full_output=True)
        oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=4)
    _lq_assert_identical(th, info, nfev, oth, oinfo, onfev)
    assert be.last_lq_refit_count() < n // 3          # the ordinary spots were fitted once


@pytest.mark.parametrize("box", [3, 7, 13])
def test_gausslq_refit_mode_tree_sums_and_a_second_fit_of_the_flagged(be, orc, box):
    """The faster mode (pmi_gausslq_set_mode(refit)): tree sums over the rows, every spot with a decision of lmdif within
    rounding distance of its threshold, a pivot tie or a nearly rank-deficient Jacobian fitted again in MINPACK's order.
    Bit-identical on (practically) every spot, every spot of these sets inside the north-star tolerance."""
    n = 20000
    spots = _lq_adversarial_spots(box, n, 900 + box)
    be.set_lq_mode("refit")
    try:
        th, info, nfev = be.gausslq_arrays(spots, full_output=True)
        refit = be.last_lq_refit_count()
    finally:
        be.set_lq_mode("strict")
    oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=orc.max_threads())
    _check_lq(th, info, nfev, oth, oinfo, onfev)
    assert 0 < refit < 0.05 * n, refit          # the second fit is the exception


def test_gausslq_edge_cases(be, orc):
    assert be.gausslq_arrays(np.zeros((0, 7, 7), np.float32)).shape == (0, 6)
    from picasso_amd import _lib
    with pytest.raises(_lib.HipBackendError, match="odd box"):
        be.gausslq_arrays(np.zeros((1, 4, 4), np.float32))
    # flat, empty and single-pixel spots: the sum <= 0 branch of the start values, zero Jacobian columns
    spots = np.zeros((6, 7, 7), np.float32)
    spots[1] += 5.0
    spots[2, 3, 3] = 100.0
    spots[3, 0, 0] = 1e6
    spots[4] = np.arange(49, dtype=np.float32).reshape(7, 7)
    spots[5] = -3.0
    th, info, nfev = be.gausslq_arrays(spots, full_output=True)
    oth, oinfo, onfev = orc.gausslq(spots, full=True)
    assert np.array_equal(info, oinfo) and np.array_equal(nfev, onfev)
    assert np.allclose(th, oth, rtol=1e-5, atol=1e-6, equal_nan=True)
    # a NaN pixel must terminate (maxfev) and not poison its neighbours
    bad = golden("gausslq_poisson7")["spots"][:3].copy()
    bad[1, 2, 2] = np.nan
    th = be.gausslq_arrays(bad)
    good = orc.gausslq(bad[[0, 2]])
    assert np.array_equal(th[[0, 2]], good)


@pytest.mark.parametrize("box", [5, 7, 9])
def test_localize_lq_pipeline_on_resident_movie(be, orc, testdata_movie, box):
    """identify -> fused cut + lmdif -> 11-column table on device vs oracle fit + host table, on the reference's own test
    movie: in the strict mode (the default) every fitted column of every row is the oracle's bit for bit (the refit mode
    of round 3 was held to 98 % identical rows here), the precision columns to float32 rounding."""
    import pandas as pd
    from picasso_amd import gausslq
    assert be.get_lq_mode() == "strict"
    for gain in (1.0, 2.0):
        cam = {"Baseline": 100.0, "Sensitivity": 0.5, "Gain": gain}
        dm = be.DeviceMovie(testdata_movie)
        try:
            t = be.localize_lq_device(dm.ptr, dm.dtype, dm.shape, box, 1500, cam)
        finally:
            dm.free()
        fr, y, x, ng = orc.identify(testdata_movie, 1500, box)
        spots = orc.get_spots(testdata_movie, fr, y, x, box, cam)
        oth = orc.gausslq(spots, threads=4)
        ids = pd.DataFrame({"frame": fr, "x": x, "y": y, "net_gradient": ng})
        ref = gausslq.locs_from_fits(ids, oth, box, em=gain > 1)
        assert list(t) == list(ref.columns)
        assert np.array_equal(t["frame"], ref["frame"].to_numpy())
        assert np.array_equal(t["net_gradient"], ref["net_gradient"].to_numpy())
        for c in ("x", "y", "photons", "sx", "sy", "bg"):
            assert np.array_equal(t[c], ref[c].to_numpy(), equal_nan=True), (c, gain)
        for c in ("lpx", "lpy", "ellipticity"):
            assert np.allclose(t[c], ref[c].to_numpy(), rtol=3e-7, atol=0, equal_nan=True), c


# ---------------------------------------------------------------------------
# render (picasso/render.py): histogram and order-preserving Gaussian
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_render_vs_goldens_and_oracle(be, orc, case):
    g = golden("render_cases")
    (y_min, x_min), (y_max, x_max) = g[case + "_viewport"]
    osamp, mbw = float(g[case + "_oversampling"]), float(g[case + "_min_blur"])
    n, hist = be.render_arrays(g["x"], g["y"], osamp, y_min, x_min, y_max, x_max)
    assert n == int(g[case + "_n"]) and np.array_equal(hist, g[case + "_hist"])
    n, img = be.render_arrays(g["x"], g["y"], osamp, y_min, x_min, y_max, x_max, g["lpx"], g["lpy"], mbw)
    ref = g[case + "_gauss_numba"]
    assert n == int(g[case + "_n"]) and img.shape == ref.shape and img.dtype == np.float32
    # same additions in the same order; only a 1-ulp float64 exp difference can survive the float32 rounding
    assert (img != ref).mean() < 1e-3
    assert np.max(np.abs(img - ref)) <= 4e-7 * float(ref.max())
    n, iso = be.render_arrays(g["x"], g["y"], osamp, y_min, x_min, y_max, x_max, g["lpx"], g["lpy"], mbw, iso=True)
    ref = g[case + "_iso_numba"]
    assert n == int(g[case + "_n"]) and (iso != ref).mean() < 1e-3 and np.max(np.abs(iso - ref)) <= 4e-7 * float(ref.max())


def test_render_large_random_vs_oracle(be, orc):
    """~2e5 localizations, oversampling 10 on a 128 px field (1280^2 image) and the undrift-style
    oversampling 1 / min_blur 1 render (every pixel receives thousands of ordered additions)."""
    rng = np.random.default_rng(5)
    N = 200_000
    x = rng.uniform(-2, 130, N).astype(np.float32)
    y = rng.uniform(-2, 130, N).astype(np.float32)
    lpx = rng.uniform(0.01, 0.15, N).astype(np.float32)
    lpy = rng.uniform(0.01, 0.15, N).astype(np.float32)
    lpx[:5] = np.nan                                        # NaN precision: np.maximum propagates, nothing is drawn
    for osamp, mbw in ((10.0, 0.0), (1.0, 1.0)):
        n, img = be.render_arrays(x, y, osamp, 0, 0, 128, 128, lpx, lpy, mbw)
        on, oimg = orc.render(x, y, osamp, [(0, 0), (128, 128)], lpx, lpy, "gaussian", mbw)
        assert n == on and img.shape == oimg.shape
        assert (img != oimg).mean() < 1e-3
        assert np.nanmax(np.abs(img - oimg)) <= 1e-6 * float(np.nanmax(oimg))
        n, hist = be.render_arrays(x, y, osamp, 0, 0, 128, 128)
        on, ohist = orc.render(x, y, osamp, [(0, 0), (128, 128)])
        assert n == on and np.array_equal(hist, ohist) and hist.sum() == n


def test_render_edge_cases(be, orc):
    e = np.zeros(0, np.float32)
    n, img = be.render_arrays(e, e, 2.0, 0, 0, 8, 8, e, e, 0.0)
    assert n == 0 and img.shape == (16, 16) and not img.any()
    n, img = be.render_arrays(e, e, 2.0, 0, 0, 8, 8)
    assert n == 0 and not img.any()
    # on the viewport border: strict inequalities (render.py:226)
    x = np.array([0.0, 8.0, 4.0, 7.99], np.float32)
    y = np.array([4.0, 4.0, 0.0, 7.99], np.float32)
    lp = np.full(4, 0.3, np.float32)
    n, img = be.render_arrays(x, y, 3.0, 0, 0, 8, 8, lp, lp, 0.0)
    on, oimg = orc.render(x, y, 3.0, [(0, 0), (8, 8)], lp, lp, "gaussian", 0.0)
    assert n == on == 1 and np.allclose(img, oimg, rtol=1e-6, atol=0)
    # a footprint wider than the image, non-square viewport
    x = np.array([5.0], np.float32); y = np.array([2.0], np.float32); lp = np.array([40.0], np.float32)
    n, img = be.render_arrays(x, y, 1.5, 0, 1, 5, 9, lp, lp, 0.0)
    on, oimg = orc.render(x, y, 1.5, [(0, 1), (5, 9)], lp, lp, "gaussian", 0.0)
    assert img.shape == oimg.shape == (8, 12) and np.allclose(img, oimg, rtol=1e-6, atol=0)


# ---------------------------------------------------------------------------
# cross-correlation / RCC pairs (picasso/imageprocess.py:27-161)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape,roi", [((64, 64), 32), ((45, 70), 32), ((33, 31), None), ((128, 96), 200), ((20, 50), 16)])
def test_xcorr_and_peak_windows_vs_numpy(be, orc, shape, roi):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    Y, X = shape
    n = 5
    yy, xx = np.mgrid[0:Y, 0:X]
    segs = np.zeros((n, Y, X))
    cy, cx = rng.uniform(0.3 * Y, 0.7 * Y, 6), rng.uniform(0.3 * X, 0.7 * X, 6)
    for s in range(n):
        dy, dx = rng.uniform(-3, 3, 2)
        for k in range(6):
            segs[s] += np.exp(-0.5 * (((yy - cy[k] - dy) / 1.3) ** 2 + ((xx - cx[k] - dx) / 1.3) ** 2))
        segs[s] += rng.uniform(0, 0.02, (Y, X))
    segs[3] = 0.0                                             # an empty segment
    xc = be.xcorr_array(segs[0], segs[1])
    ref = orc.xcorr(segs[0], segs[1])
    assert xc.shape == ref.shape and np.max(np.abs(xc - ref)) < 1e-12 * np.abs(ref).max()
    peak, valid, rois, (Y_, X_) = be.rcc_pairs_arrays(segs, roi, 5)
    p = 0
    for i in range(n - 1):
        for j in range(i + 1, n):
            want = orc.peak_window(segs[i], segs[j], 5, roi)
            if want is None:
                assert valid[p] == -1
            else:
                ym, xm, wy, wx, win = want
                assert (Y_, X_) == (wy, wx) and tuple(peak[p]) == (ym, xm)
                assert (valid[p] == 1) == (win is not None)
                if win is not None:
                    assert np.max(np.abs(rois[p] - win)) < 1e-12 * np.abs(ref).max()
            p += 1
    # an explicit pair list (the share of one rank in dist.rcc_sharded) equals the same rows of the full run
    allp = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    sub = allp[1::3]
    pk2, va2, ro2, crop2 = be.rcc_pairs_arrays(segs, roi, 5, pairs=sub)
    assert crop2 == (Y_, X_) and np.array_equal(pk2, peak[1::3]) and np.array_equal(va2, valid[1::3])
    assert np.array_equal(ro2, rois[1::3])


def test_photon_conversion_division_is_correctly_rounded(be, orc):
    """The kernels divide by the (uniform) gain with a 3-instruction Markstein sequence instead of an IEEE
    division; get_spots must stay bit-identical to float32 (x - baseline) * sensitivity / gain for any gain."""
    rng = np.random.default_rng(99)
    for dtype in (np.uint16, np.float32):
        if dtype == np.uint16:
            mov = rng.integers(0, 65535, size=(3, 40, 48)).astype(np.uint16)
        else:
            mov = (rng.uniform(-50, 70000, size=(3, 40, 48)) * rng.choice([1e-6, 1e-3, 1.0, 1e3], size=(3, 40, 48))).astype(np.float32)
        n = 400
        fr = rng.integers(0, 3, n).astype(np.int32); y = rng.integers(3, 37, n).astype(np.int32); x = rng.integers(3, 45, n).astype(np.int32)
        for gain in (1.0, 2.0, 0.37, 3.3, 55.1, 300.0, 1e-3, 7.0 / 3.0, 1e20, float(np.float32(1.0000001))):
            for sens in (1.0, 0.123, 4.7):
                cam = {"Baseline": 99.5, "Sensitivity": sens, "Gain": gain}
                a = be.get_spots_array(mov, fr, y, x, 7, cam["Baseline"], cam["Sensitivity"], cam["Gain"])
                raw = np.stack([mov[f, yy - 3:yy + 4, xx - 3:xx + 4] for f, yy, xx in zip(fr, y, x)]).astype(np.float32)
                want = (raw - np.float32(99.5)) * np.float32(sens) / np.float32(gain)
                assert want.dtype == np.float32 and np.array_equal(a, want), (dtype, gain, sens)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,box", [((2, 300, 600), 3), ((1, 1100, 1040), 5), ((3, 70, 2100), 3)])
def test_identify_dense_frames_order(be, orc, shape, box):
    """Noise frames with a threshold that keeps every maximum: thousands of rows per frame, more than one LDS
    tile of sort keys (4096) and several sort blocks per frame; rows still come back in (frame, y, x) order
    with the oracle's values."""
    rng = np.random.default_rng(shape[1] * 7 + box)
    mov = rng.integers(100, 4000, size=shape).astype(np.uint16)
    a = be.identify_arrays(mov, -1e9, box)
    b = orc.identify(mov, -1e9, box, threads=4)
    assert len(a[0]) == len(b[0]) > 4096 * shape[0] // 2
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    key = (a[0].astype(np.int64) << 40) | (a[1].astype(np.int64) << 20) | a[2].astype(np.int64)
    assert np.all(np.diff(key) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("lq", [False, True])
def test_fused_pipeline_capacity_overflow_is_clean(be, lq):
    """A table capacity below the row count: the fused submission reports the count, fits nothing, leaves the
    table alone (no stale identification is followed into the movie), and the resubmission with room gives
    the same rows as a roomy first call."""
    import ctypes
    from picasso_amd import _lib
    rng = np.random.default_rng(3)
    mov = rng.poisson(40, size=(6, 96, 160)).astype(np.uint16) + 100
    for f in range(6):
        for _ in range(40):
            y, x = rng.integers(6, 90), rng.integers(6, 154)
            mov[f, y - 1:y + 2, x - 1:x + 2] += 300
            mov[f, y, x] += 900
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    dm = be.DeviceMovie(mov)
    fn = be.localize_lq_device if lq else be.localize_mle_device
    roomy = fn(dm.ptr, dm.dtype, dm.shape, 7, 500.0, cam, cap=100000)
    n = len(roomy["frame"])
    assert n > 100
    L = _lib.load()
    ncol = _lib.PMI_LQ_COLUMNS if lq else _lib.PMI_LOC_COLUMNS
    cap = 16
    table, dn = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(L.pmi_malloc(ctypes.byref(table), ncol * cap * 4), "malloc")
    _lib.check(L.pmi_malloc(ctypes.byref(dn), 8), "malloc")
    fill = np.full(ncol * cap, 0x5A5A5A5A, np.uint32)
    _lib.check(L.pmi_memcpy_h2d(table, _lib.ptr(fill), fill.nbytes), "h2d")
    if lq:
        rc = L.pmi_localize_lq_dev(dm.ptr, 0, 6, 96, 160, 7, 500.0, None, 0, 5, 100.0, 1.0, 1.0, 0, table, cap, dn, None)
    else:
        rc = L.pmi_localize_mle_dev(dm.ptr, 0, 6, 96, 160, 7, 500.0, None, 0, 5, 100.0, 1.0, 1.0, 1e-3, 100, 1, table, cap, dn, None)
    _lib.check(rc, "fused")
    _lib.check(L.pmi_stream_synchronize(None), "sync")
    got_n = np.zeros(1, np.int64)
    back = np.zeros_like(fill)
    _lib.check(L.pmi_memcpy_d2h(_lib.ptr(got_n), dn, 8), "d2h")
    _lib.check(L.pmi_memcpy_d2h(_lib.ptr(back), table, back.nbytes), "d2h")
    assert int(got_n[0]) == n and np.array_equal(back, fill)
    L.pmi_free(table); L.pmi_free(dn)
    retried = fn(dm.ptr, dm.dtype, dm.shape, 7, 500.0, cam, cap=16)      # grows to the reported count
    for c in roomy:
        assert np.array_equal(roomy[c], retried[c], equal_nan=True), c
    dm.free()


def test_peak_fit_device_vs_oracle(be, orc):
    """The bounded Gaussian fit of the correlation window (scipy curve_fit in the reference,
    picasso/imageprocess.py:129-135) on the device against the oracle's restatement, which is pinned against
    scipy itself in test_oracle_golden.py: centres within 1e-7 px, same termination status."""
    rng = np.random.default_rng(3)
    for box in (5, 7, 9):
        h = box // 2
        y, x = np.mgrid[-h:h + 1, -h:h + 1]
        rois = []
        for t in range(200):
            a = rng.uniform(5, 500); xc, yc = rng.uniform(-0.7, 0.7, 2); s = rng.uniform(0.6, 2.5)
            b = rng.uniform(0, 50) if t % 3 else 0.0
            roi = a * np.exp(-0.5 * ((x - xc) ** 2 + (y - yc) ** 2) / s ** 2) + b + rng.normal(0, 0.02 * a, (box, box))
            rois.append(np.abs(roi) if t % 2 else np.maximum(roi, 0.0))
        rois = np.array(rois)
        popt, status = be.peak_fit_arrays(rois)
        for k in range(len(rois)):
            o, ost, _ = orc.peak_fit(rois[k])
            assert status[k] == ost, (box, k, status[k], ost)
            assert np.max(np.abs(popt[k, 1:3] - o[1:3])) < 1e-7 and abs(popt[k, 3] - o[3]) < 1e-6 * max(1.0, o[3])
    neg = rois[:2].copy(); neg[1] -= neg[1].max()
    popt, status = be.peak_fit_arrays(neg)
    assert status[0] > 0 and status[1] == -2             # a negative window minimum: curve_fit raises, so does the wrapper
    bad = rois[:3].copy(); bad[1, 2, 2] = np.nan; bad[2, 0, 0] = np.inf
    popt, status = be.peak_fit_arrays(bad)
    assert status[0] > 0 and status[1] == -3 and status[2] == -3      # curve_fit(check_finite=True): ValueError


def test_get_image_shift_raises_what_curve_fit_raises(be):
    """picasso/imageprocess.py:129-135 calls scipy's curve_fit, which raises ValueError on non-finite input and
    RuntimeError when the fit runs out of function evaluations; the mirror raises the same (checked against scipy)."""
    from scipy.optimize import curve_fit

    from picasso_amd import imageprocess
    rng = np.random.default_rng(9)
    a = rng.poisson(5.0, (64, 64)).astype(np.float64)

    def model(xy, A, xc, yc, s, bb):
        return A * np.exp(-0.5 * ((xy[0] - xc) ** 2 + (xy[1] - yc) ** 2) / s ** 2) + bb
    y, x = np.mgrid[-2:3, -2:3]
    win = np.zeros((5, 5)); win[2, 2] = np.nan
    with pytest.raises(ValueError):         # scipy's own verdict on a window that holds a NaN
        curve_fit(model, (x.ravel(), y.ravel()), win.ravel(), p0=[1, 0, 0, 1, 0], bounds=([0, -np.inf, -np.inf, 0, 0], np.inf))
    popt, status = be.peak_fit_arrays(win[None])
    assert status[0] == -3                  # ... and the device's: status -3
    orig = be.rcc_shifts_arrays
    try:                                    # the fit statuses of pmi_rcc_shifts -> curve_fit's exceptions
        be.rcc_shifts_arrays = lambda *args, **kw: (np.zeros((1, 2)), np.array([-3], np.int32))
        with pytest.raises(ValueError, match="infs or NaNs"):
            imageprocess.get_image_shift(a, a, 5, 32)
        be.rcc_shifts_arrays = lambda *args, **kw: (np.zeros((1, 2)), np.array([0], np.int32))
        with pytest.raises(RuntimeError, match="Optimal parameters not found"):
            imageprocess.get_image_shift(a, a, 5, 32)
    finally:
        be.rcc_shifts_arrays = orig


def test_rcc_shifts_on_device_match_reference_goldens(be):
    """get_image_shift end to end on the device (correlation, crop, peak, window, fit) against the shifts the
    reference computed (tests/golden/undrift_rcc.npz): 1e-6 px."""
    from picasso_amd import imageprocess
    g = golden("undrift_rcc")
    seg = g["segments"]
    shifts, status = be.rcc_shifts_arrays(seg, 32, 5)
    p = 0
    for i in range(len(seg) - 1):
        for j in range(i + 1, len(seg)):
            assert status[p] > 0
            assert abs(shifts[p, 0] - g["raw_shift_y"][i, j]) < 1e-6 and abs(shifts[p, 1] - g["raw_shift_x"][i, j]) < 1e-6
            p += 1
    sy, sx = imageprocess.rcc(list(seg), 32)
    assert np.max(np.abs(sy - g["shift_y"])) < 1e-6 and np.max(np.abs(sx - g["shift_x"])) < 1e-6
    empty = seg.copy(); empty[1] = 0
    shifts, status = be.rcc_shifts_arrays(empty, 32, 5, [(0, 1), (0, 2)])
    assert status[0] == -1 and np.all(shifts[0] == 0) and status[1] > 0
