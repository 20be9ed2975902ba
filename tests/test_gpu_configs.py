"""GPU tier: the workloads BASELINE.json names, end to end.

config 2 at FULL size (10k frames, 512x512, ~1e6 spots): every identification and every fit against the
oracle (its C restatement runs on all host threads of the GPU box in seconds), plus size-independent
properties — ordering, determinism, shard consistency (frames shard without a halo), agreement with the
simulated ground truth; config 3 (gausslq + Gaussian render at oversampling 10, 2000 frames), config 4's
geometry in miniature and config 5 (13x13 astigmatic MLE + zfit) against the oracle composition.
"""
import ctypes

import numpy as np
import pandas as pd
import pytest

from conftest import assert_mle_rows, golden

pytestmark = pytest.mark.gpu
CAM = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}


@pytest.fixture(scope="module")
def be():
    from picasso_amd import backend
    return backend


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _localize_resident(be, movie_t, box=7, min_ng=5000.0, f_lo=None, f_hi=None, lq=False):
    F, H, W = movie_t.shape
    fn = be.localize_lq_device if lq else be.localize_mle_device
    return fn(ctypes.c_void_p(movie_t.data_ptr()), np.uint16, (F, H, W), box, min_ng, CAM, f_lo=f_lo, f_hi=f_hi)


def test_config2_full_size_properties(be, orc):
    import torch
    from picasso_amd import synth
    F = 10000
    movie, truth = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda", return_truth=True)
    torch.cuda.synchronize()
    t = _localize_resident(be, movie)
    n = len(t["frame"])
    assert 0.9e6 < n < 1.2e6
    # ordered by frame; every frame contributes; dtypes of the 17-column table
    fr = t["frame"].astype(np.int64)
    assert np.all(np.diff(fr) >= 0) and fr[0] == 0 and fr[-1] == F - 1
    assert t["frame"].dtype == np.uint32 and t["iterations"].dtype == np.uint32 and t["x"].dtype == np.float32
    # idempotent / deterministic: a second pass gives the same bits
    t2 = _localize_resident(be, movie)
    for c in t:
        assert np.array_equal(t[c], t2[c], equal_nan=True), c
    # frames shard without a halo: two half ranges concatenate to the whole table
    a = _localize_resident(be, movie, f_lo=0, f_hi=F // 2 - 1)
    b = _localize_resident(be, movie, f_lo=F // 2, f_hi=F - 1)
    for c in t:
        assert np.array_equal(np.concatenate([a[c], b[c]]), t[c], equal_nan=True), c
    # identical spot indices at FULL size: every identification of the 10 000 frames against the oracle
    host = movie.cpu().numpy()
    ofr, oy, ox, ong = orc.identify(host, 5000.0, 7, threads=orc.max_threads())
    assert len(ofr) == n and np.array_equal(ofr, fr) and np.array_equal(ong, t["net_gradient"])
    idf = be.identify_arrays(host[:3000], 5000.0, 7)          # the host-buffer entry point on the first 1.5 GB
    k = int(np.searchsorted(ofr, 3000))
    assert np.array_equal(idf[0], ofr[:k]) and np.array_equal(idf[1], oy[:k]) and np.array_equal(idf[2], ox[:k])
    # ... and the fit of every one of them within the north-star tolerance (x, y, sigma 1e-3 px, photons 1e-2)
    spots = orc.get_spots(host, ofr, oy, ox, 7, CAM)
    th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=orc.max_threads())
    # ALL rows: identical iteration counts, x / y / sigma < 1e-3 px, photons < 1e-2 (borderline convergence
    # decisions are re-fitted on the device in the reference's arithmetic, inside the same call)
    assert_mle_rows(t["x"], t["y"], t["sx"], t["sy"], t["photons"], t["iterations"],
                    th[:, 0] + ox - 3, th[:, 1] + oy - 3, th[:, 4], th[:, 5], th[:, 2], it, label="config 2, all rows")
    refit = be.last_refit_count()
    assert 0 < refit < 0.05 * n          # the second fit is the exception (0.7 % of the spots here)
    del host, idf, spots
    # a slice of frames against the oracle: identical identification set, fit within tolerance
    sl = slice(4000, 4040)
    sub = movie[sl].cpu().numpy()
    ofr, oy, ox, ong = orc.identify(sub, 5000.0, 7)
    m = (fr >= sl.start) & (fr < sl.stop)
    assert m.sum() == len(ofr)
    assert np.array_equal(t["net_gradient"][m], ong) and np.array_equal(fr[m] - sl.start, ofr)
    spots = orc.get_spots(sub, ofr, oy, ox, 7, CAM)
    th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=4)
    assert_mle_rows(t["x"][m], t["y"][m], t["sx"][m], t["sy"][m], t["photons"][m], t["iterations"][m],
                    th[:, 0] + ox - 3, th[:, 1] + oy - 3, th[:, 4], th[:, 5], th[:, 2], it, label="frame slice")
    assert np.max(np.abs(t["photons"][m] - th[:, 2]) / th[:, 2]) < 1e-4
    # agreement with the simulation: every localization sits on a simulated emitter of its frame
    tx, ty, tf = truth["x"], truth["y"], truth["frame"]
    fsel = (tf >= sl.start) & (tf < sl.stop)
    for f in range(sl.start, sl.stop, 7):
        li = np.nonzero(fr == f)[0]
        ti = np.nonzero(fsel & (tf == f))[0]
        d = np.hypot(t["x"][li][:, None] - tx[ti][None, :], t["y"][li][:, None] - ty[ti][None, :])
        assert np.all(d.min(axis=1) < 0.3)
    del movie


def test_fused_pipeline_with_every_spot_in_the_reference_arithmetic(be, orc):
    """`value_strict` of the bench line: pmi_localize_mle_dev in MLE mode `strict` — identify, then the start-value kernel and
    the refilling Newton kernel of csrc/gaussmle_strict.hip reading the MOVIE, then the table — on 1500 frames: the
    identifications are the oracle's and photons / bg / sx / sy / iterations of every row its bits (x and y: theta + the
    pixel position in float32)."""
    import torch
    from picasso_amd import synth
    F = 1500
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda", seed=4242)
    torch.cuda.synchronize()
    be.set_mle_mode("strict")
    try:
        t = _localize_resident(be, movie)
    finally:
        be.set_mle_mode("refit")
    host = movie.cpu().numpy()
    fr, y, x, ng = orc.identify(host, 5000.0, 7, threads=orc.max_threads())
    assert len(t["frame"]) == len(fr) > 1.2e5 and np.array_equal(t["frame"], fr.astype(np.uint32)) and np.array_equal(t["net_gradient"], ng)
    spots = orc.get_spots(host, fr, y, x, 7, CAM)
    th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=orc.max_threads())
    assert np.array_equal(t["iterations"], it.astype(np.uint32))
    for c, k in (("photons", 2), ("bg", 3), ("sx", 4), ("sy", 5)):
        assert np.array_equal(t[c].view(np.uint32), th[:, k].view(np.uint32)), c
    assert np.array_equal(t["x"], (th[:, 0] + x - 3).astype(np.float32)) or np.max(np.abs(t["x"] - (th[:, 0] + x - 3))) < 1e-5
    assert np.array_equal(t["y"], (th[:, 1] + y - 3).astype(np.float32)) or np.max(np.abs(t["y"] - (th[:, 1] + y - 3))) < 1e-5


def test_config3_gausslq_and_render(be, orc):
    """gausslq path + Gaussian render at oversampling 10 (BASELINE.json configs[2]) at its FULL size — 10 000
    frames, 1e6 spots — vs the oracle composition (the C restatement of MINPACK lmdif fits them in ~10 s on the
    box's host threads)."""
    import torch
    from picasso_amd import gausslq, synth
    F = 10000
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda", seed=77)
    torch.cuda.synchronize()
    t = _localize_resident(be, movie, lq=True)
    sub = movie.cpu().numpy()
    fr, y, x, ng = orc.identify(sub, 5000.0, 7, threads=orc.max_threads())
    assert len(t["frame"]) == len(fr) > 900000 and np.array_equal(t["net_gradient"], ng)
    spots = orc.get_spots(sub, fr, y, x, 7, CAM)
    oth = orc.gausslq(spots, threads=orc.max_threads())
    ref = gausslq.locs_from_fits(pd.DataFrame({"frame": fr, "x": x, "y": y, "net_gradient": ng}), oth, 7, em=False)
    ref = ref.sort_index()          # the reference's quicksort by frame is not stable; undo it (identification order)
    # the strict mode (the default) is MINPACK's own arithmetic: every fitted column of every one of the 1e6 rows is the
    # oracle's bit for bit (the refit mode of round 3 was held to 98 % identical rows and 1e-3 px here)
    assert be.get_lq_mode() == "strict"
    for c in ("x", "y", "photons", "bg", "sx", "sy"):
        want = ref[c].to_numpy()
        assert np.array_equal(t[c], want, equal_nan=True), (c, int((t[c] != want).sum()))
    for c in ("lpx", "lpy"):          # Mortensen's precision from those columns (float32 expression order of the kernel): 1e-6 relative
        want = ref[c].to_numpy()
        assert np.allclose(t[c], want, rtol=1e-6, atol=0, equal_nan=True), c
    # render what the GPU localized, against the oracle render of the same table
    n, img = be.render_arrays(t["x"], t["y"], 10.0, 0, 0, 512, 512, t["lpx"], t["lpy"], 0.0)
    on, oimg = orc.render(t["x"], t["y"], 10.0, [(0, 0), (512, 512)], t["lpx"], t["lpy"], "gaussian", 0.0)
    assert n == on == len(t["x"]) and img.shape == (5120, 5120)
    assert (img != oimg).mean() < 1e-4 and np.max(np.abs(img - oimg)) <= 1e-6 * float(oimg.max())
    assert abs(float(img.sum(dtype=np.float64)) - n) < 0.02 * n          # mass conservation (tests/test_render.py)


def test_config5_astigmatic_13x13_and_zfit(be, orc):
    """13x13 ROI astigmatic MLE fit + zfit (BASELINE.json configs[4]) vs the oracle."""
    from math import erf, sqrt
    g = golden("zfit_calib3d")
    cx, cy = g["cx"], g["cy"]
    rng = np.random.default_rng(13)
    N, box, c = 1500, 13, 6
    z = rng.uniform(-400, 400, N)
    sx = np.polyval(cx, z); sy = np.polyval(cy, z)
    idx = np.arange(box)
    spots = np.empty((N, box, box), np.float32)
    for i in range(N):
        x0, y0 = c + rng.uniform(-0.5, 0.5, 2)
        ex = np.array([0.5 * (erf((k - x0 + .5) / (sqrt(2) * sx[i])) - erf((k - x0 - .5) / (sqrt(2) * sx[i]))) for k in idx])
        ey = np.array([0.5 * (erf((k - y0 + .5) / (sqrt(2) * sy[i])) - erf((k - y0 - .5) / (sqrt(2) * sy[i]))) for k in idx])
        spots[i] = rng.poisson(rng.uniform(3000, 9000) * np.outer(ey, ex) + rng.uniform(5, 25))
    th, cr, ll, it = be.gaussmle_arrays(spots, 1e-3, 100, "sigmaxy")
    oth, ocr, oll, oit = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=4)
    assert_mle_rows(th[:, 0], th[:, 1], th[:, 4], th[:, 5], th[:, 2], it,
                    oth[:, 0], oth[:, 1], oth[:, 4], oth[:, 5], oth[:, 2], oit, label="config 5 13x13")
    assert np.max(np.abs(th[:, 2] - oth[:, 2]) / oth[:, 2]) < 1e-4
    zz, sq = be.zfit_arrays(th[:, 4], th[:, 5], cx, cy)
    oz, osq = orc.zfit(th[:, 4], th[:, 5], cx, cy, threads=4)
    assert np.max(np.abs(zz - oz)) < 5e-5 and np.max(np.abs(sq - osq)) < 1e-9
    # the fitted z follows the simulated one (astigmatism calibration is monotonic in this range)
    assert np.median(np.abs(zz - z)) < 25.0


def test_config5_fused_pipeline_on_the_astigmatic_movie(be, orc):
    """Config 5 through the path the benchmark times — pmi_localize_mle_dev with box 13 on the astigmatic movie, then
    zfit — against the oracle on every row, at 3000 frames (~3e5 spots; tools/parity_config5.py runs the same comparison
    at the full 50 000 frames, profiles/r03_parity_config5.json, r05_parity_config5.json)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_config5
    out = parity_config5.run(F=3000, chunk=1000)
    assert out["rows_compared"] == out["localizations_gpu"] > 2.5e5 and out["chunks_with_identification_mismatch"] == 0
    assert out["rows_with_different_iterations"] == 0
    w = out["max_abs_diff_rows_below_max_it"]
    assert max(w["x"], w["y"], w["sx"], w["sy"]) < 1e-3 and w["photons_rel"] < 1e-2 and w["z"] < 0.05 and w["lpx_rel"] < 2e-3          # z in nm over a +-400 nm range: 2e-6 px of width times the slope of the calibration
    assert 0 < out["refit_spots"] < 0.05 * out["localizations_gpu"]


def test_config5_lq3d_twin_on_the_astigmatic_movie(be, orc):
    """Config 5 through the reference's 3-D DEFAULT route (picasso/zfit.py:300,472: fitting_method="gausslq"):
    pmi_localize_lq_dev with box 13 on 3000 frames of the astigmatic movie, then the z fit, against the oracle's identify ->
    get_spots -> lmdif -> table -> zfit on every row.  The strict mode is MINPACK's own arithmetic: theta is the oracle's bit
    for bit, so the table columns are compared for equality, z to the tolerance of the Brent search."""
    import torch
    from picasso_amd import synth
    g = golden("zfit_calib3d")
    F = 3000
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda", sigma=(1.1, 2.4), astigmatic=True,
                                 photons=(3000.0, 9000.0), seed=synth.DEFAULT_SEED + 5)
    torch.cuda.synchronize()
    assert be.get_lq_mode() == "strict"
    t = be.localize_lq_device(movie.data_ptr(), np.uint16, tuple(movie.shape), 13, 5000.0, CAM)
    n = len(t["frame"])
    assert n > 2.5e5
    host = movie.cpu().numpy()
    del movie
    fr, y, x, ng = orc.identify(host, 5000.0, 13, threads=orc.max_threads())
    assert n == len(fr) and np.array_equal(t["frame"], fr.astype(np.uint32)) and np.array_equal(t["net_gradient"], ng)
    spots = orc.get_spots(host, fr, y, x, 13, CAM)
    th = orc.gausslq(spots, threads=orc.max_threads())
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)      # noqa: E731
    assert same(t["x"], (th[:, 0].astype(np.float64) + x).astype(np.float32)) and same(t["y"], (th[:, 1].astype(np.float64) + y).astype(np.float32))
    assert same(t["photons"], th[:, 2]) and same(t["bg"], th[:, 3]) and same(t["sx"], th[:, 4]) and same(t["sy"], th[:, 5])
    z, sq = be.zfit_arrays(t["sx"], t["sy"], g["cx"], g["cy"])
    oz, osq = orc.zfit(th[:, 4], th[:, 5], g["cx"], g["cy"], threads=orc.max_threads())
    ok = np.isfinite(oz)
    assert np.array_equal(np.isfinite(z), ok) and np.max(np.abs(z[ok] - oz[ok])) < 5e-5
    assert be.last_lq_refit_count() < 1e-3 * n


def test_config4_geometry_shard_in_miniature(be, orc):
    """Config 4's frame geometry (2048 x 2048, ~1600 spots per frame) at a length the oracle can follow: the
    identification set and net gradients are the oracle's, the fit is within tolerance, frame ranges concatenate
    to the whole table (what the 8-rank form relies on), and RCC undrift runs on the table.  The full 25 000-frame
    share of one rank is measured by tools/bench_config4_shard.py (profiles/r01g_config4_shard.jsonl)."""
    import torch
    from picasso_amd import postprocess, synth
    F, H, W = 48, 2048, 2048
    movie = synth.simulate_movie(F, H, W, emitters_per_frame=1856, device="cuda", chunk_frames=16)
    torch.cuda.synchronize()
    t = _localize_resident(be, movie)
    fr = t["frame"].astype(np.int64)
    per = np.bincount(fr, minlength=F)
    assert per.min() > 1400 and per.max() < 1800 and np.all(np.diff(fr) >= 0)
    parts = [_localize_resident(be, movie, f_lo=a, f_hi=b) for a, b in ((0, 5), (6, 23), (24, 47))]
    for c in t:
        assert np.array_equal(np.concatenate([p[c] for p in parts]), t[c], equal_nan=True), c
    sub = movie[10:13].cpu().numpy()
    ofr, oy, ox, ong = orc.identify(sub, 5000.0, 7, threads=4)
    m = (fr >= 10) & (fr < 13)
    assert m.sum() == len(ofr) and np.array_equal(fr[m] - 10, ofr) and np.array_equal(t["net_gradient"][m], ong)
    spots = orc.get_spots(sub, ofr, oy, ox, 7, CAM)
    th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=4)
    assert_mle_rows(t["x"][m], t["y"][m], t["sx"][m], t["sy"][m], t["photons"][m], t["iterations"][m],
                    th[:, 0] + ox - 3, th[:, 1] + oy - 3, th[:, 4], th[:, 5], th[:, 2], it, label="config 4 miniature")
    locs = pd.DataFrame(t)
    drift, und = postprocess.undrift(locs, [{"Frames": F, "Height": H, "Width": W}, {"Pixelsize": 130}], 8, display=False)
    assert len(und) == len(locs) and len(drift) == F and float(np.abs(drift.to_numpy()).max()) < 0.2
    del movie


def test_two_pipelines_in_flight_give_the_rows_of_one_pass():
    """pmi_scratch_bank: two frame ranges queued on two streams, each with its own scratch bank (the fit of range A
    runs beside the scan of range B), produce — concatenated — exactly the table of one pass over all frames."""
    import torch
    from picasso_amd import _lib, synth
    L = _lib.load()
    F, half = 400, 200
    movie = synth.simulate_movie(F, 256, 256, emitters_per_frame=40, seed=11, device="cuda")
    torch.cuda.synchronize()
    cap = 200 * F
    px = 256 * 256 * 2

    def run(ptr, frames, table, d_n, stream, bank):
        _lib.check(L.pmi_scratch_bank(bank), "pmi_scratch_bank")
        rc = L.pmi_localize_mle_dev(ctypes.c_void_p(ptr), 0, frames, 256, 256, 7, 5000.0, None, 0, frames - 1, 100.0, 1.0, 1.0,
                                    1e-3, 100, _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(table.data_ptr()), cap,
                                    ctypes.c_void_p(d_n.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
        _lib.check(rc, "pmi_localize_mle_dev")

    try:
        whole = torch.zeros((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda")
        n_whole = torch.zeros(1, dtype=torch.int64, device="cuda")
        run(movie.data_ptr(), F, whole, n_whole, torch.cuda.current_stream(), 0)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        parts = [torch.zeros((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda") for _ in range(2)]
        ns = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(2)]
        for rep in range(3):                  # repeated: the two pipelines really overlap from the second pass on
            for k in range(2):
                run(movie.data_ptr() + k * half * px, half, parts[k], ns[k], streams[k], k)
        torch.cuda.synchronize()
    finally:
        _lib.check(L.pmi_scratch_bank(0), "pmi_scratch_bank")
    n = int(n_whole.item())
    na, nb = int(ns[0].item()), int(ns[1].item())
    assert n > 10000 and na + nb == n
    w = whole[:, :n].cpu().numpy()
    a, b = parts[0][:, :na].cpu().numpy().copy(), parts[1][:, :nb].cpu().numpy().copy()
    b[0] += half                              # column 0 = frame (uint32 cells): range B counts from its own first frame
    both = np.concatenate([a, b], axis=1)
    assert np.array_equal(both, w)
    assert L.pmi_scratch_bank(2) != 0         # only banks 0 and 1 exist


def test_fused_call_keeps_two_frame_ranges_in_flight():
    """pmi_localize_mle_dev cuts a large frame range in two and runs the scan of the second half beside the fit of the
    first (its own side stream, the inner scratch bank): the table is the one of a single pass, bit for bit — whole
    movie, a frame range with an odd number of frames, an ROI; the re-fit counts add up; a capacity below the sum of
    the two halves reports the sum and leaves the table untouched."""
    import torch
    from picasso_amd import _lib, backend, synth
    L = _lib.load()
    F = 1100                                    # 2.9e8 pixels: above the threshold of the two-range schedule
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=60, seed=23, device="cuda")
    torch.cuda.synchronize()
    cap = 150 * F

    def run(ranges, f_lo, f_hi, roi, cap_, fill=None):
        _lib.check(L.pmi_localize_set_ranges(ranges), "pmi_localize_set_ranges")
        table = torch.zeros((_lib.PMI_LOC_COLUMNS, cap_), dtype=torch.int32, device="cuda")
        if fill is not None:
            table.fill_(fill)
        d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
        r = (ctypes.c_int64 * 4)(*roi) if roi else None
        rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, 7, 5000.0, r, f_lo, f_hi, 100.0, 1.0, 1.0,
                                    1e-3, 100, _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(table.data_ptr()), cap_,
                                    ctypes.c_void_p(d_n.data_ptr()), None)
        _lib.check(rc, "pmi_localize_mle_dev")
        torch.cuda.synchronize()
        return table, int(d_n.item()), backend.last_refit_count()

    try:
        n_full = 0
        for f_lo, f_hi, roi in ((0, F - 1, None), (3, F - 5, None), (0, F - 1, (10, 20, 500, 490))):
            one, n1, refit1 = run(1, f_lo, f_hi, roi, cap)
            n_full = n_full or n1
            for rep in range(2):
                two, n2, refit2 = run(2, f_lo, f_hi, roi, cap)
                assert n1 == n2 and n1 > 20000, (n1, n2)
                assert torch.equal(one[:, :n1], two[:, :n2]), (f_lo, f_hi, roi, rep)
                assert refit1 == refit2 and refit1 > 0
        # the pixel hand-off from the scan's exact stage to the fit (off by default): the same table, bit for bit
        _lib.check(L.pmi_localize_set_handoff(1), "pmi_localize_set_handoff")
        try:
            hand = {}
            for ranges in (1, 2):
                hand[ranges] = run(ranges, 0, F - 1, None, cap)[:2]
                assert hand[ranges][1] == n_full
            hroi = run(1, 0, F - 1, (10, 21, 500, 490), cap)[:2]          # an ROI that starts off an 8-pixel boundary
            _lib.check(L.pmi_localize_set_handoff(0), "pmi_localize_set_handoff")
            for ranges in (1, 2):
                plain, npl, _ = run(ranges, 0, F - 1, None, cap)
                h, nh = hand[ranges]
                assert npl == nh and torch.equal(plain[:, :npl], h[:, :nh]), ranges
            plain, npl, _ = run(1, 0, F - 1, (10, 21, 500, 490), cap)
            assert npl == hroi[1] and torch.equal(plain[:, :npl], hroi[0][:, :npl])
        finally:
            _lib.check(L.pmi_localize_set_handoff(0), "pmi_localize_set_handoff")
        # capacity between the first half's count and the total: nothing may be written; the count reported is the rows
        # needed — or, when even the candidates of the deferred exact stage overflow their scratch, their number (an upper bound)
        small, n_small, _ = run(2, 0, F - 1, None, int(n_full * 0.75), fill=0x5A5A5A5A)
        assert n_full <= n_small < 1.5 * n_full and bool((small == 0x5A5A5A5A).all())
        _lib.check(L.pmi_localize_set_defer(0), "pmi_localize_set_defer")
        small, n_small, _ = run(2, 0, F - 1, None, int(n_full * 0.75), fill=0x5A5A5A5A)
        assert n_small == n_full and bool((small == 0x5A5A5A5A).all())
        assert L.pmi_localize_set_ranges(3) != 0
    finally:
        _lib.check(L.pmi_localize_set_ranges(2), "pmi_localize_set_ranges")
        _lib.check(L.pmi_localize_set_defer(1), "pmi_localize_set_defer")


def test_fused_least_squares_call_keeps_two_frame_ranges_in_flight():
    """pmi_localize_lq_dev (round 4: nothing in it waits for the stream any more) cuts a large frame range in two like the
    MLE call: the same 11-column table, bit for bit, with one range and with two; a capacity below the sum reports the sum
    and leaves the table untouched."""
    import torch
    from picasso_amd import _lib, synth
    L = _lib.load()
    F = 1100
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=60, seed=29, device="cuda")
    torch.cuda.synchronize()

    def run(ranges, f_lo, f_hi, roi, cap_, fill=0):
        _lib.check(L.pmi_localize_set_ranges(ranges), "pmi_localize_set_ranges")
        table = torch.full((_lib.PMI_LQ_COLUMNS, cap_), fill, dtype=torch.int32, device="cuda")
        d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
        r = (ctypes.c_int64 * 4)(*roi) if roi else None
        rc = L.pmi_localize_lq_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, 512, 512, 7, 5000.0, r, f_lo, f_hi, 100.0, 1.0, 1.0, 0,
                                   ctypes.c_void_p(table.data_ptr()), cap_, ctypes.c_void_p(d_n.data_ptr()), None)
        _lib.check(rc, "pmi_localize_lq_dev")
        torch.cuda.synchronize()
        return table, int(d_n.item())

    try:
        cap = 150 * F
        for f_lo, f_hi, roi in ((0, F - 1, None), (3, F - 5, (10, 20, 500, 490))):
            one, n1 = run(1, f_lo, f_hi, roi, cap)
            for rep in range(2):
                two, n2 = run(2, f_lo, f_hi, roi, cap)
                assert n1 == n2 and n1 > 20000, (n1, n2)
                assert torch.equal(one[:, :n1], two[:, :n2]), (f_lo, f_hi, roi, rep)
        full, n_full = run(1, 0, F - 1, None, cap)
        small, n_small = run(2, 0, F - 1, None, int(n_full * 0.75), fill=0x5A5A5A5A)
        assert n_small == n_full and bool((small == 0x5A5A5A5A).all())
    finally:
        _lib.check(L.pmi_localize_set_ranges(2), "pmi_localize_set_ranges")


@pytest.mark.parametrize("case", ["u16_box7", "u16_box5_roi", "u16_box9", "u16_box13", "u8_box7", "i16_box7_roi", "low_threshold", "tight_cap"])
def test_fused_call_with_the_exact_stage_of_identify_in_the_fit(case):
    """pmi_localize_set_defer: the packed scan emits candidates and the fit's start-value kernel evaluates the float32 net
    gradient in the reference's order, the first-argmax rule and the threshold from the rows it reads anyway
    (picasso/localize.py:97-134, 202-244, 288).  The table is the one of the scan's own exact stage, bit for bit: boxes
    of both group sizes, an ROI that starts off an 8-pixel boundary (maxima in the crop's row / column H: their stencils wrap
    to its last row / column), uint8 and int16 movies, a threshold so low that the waves keep deciding their candidates
    themselves, and a capacity with no room for rejected candidates."""
    import torch
    from picasso_amd import _lib, backend, synth
    L = _lib.load()
    F = 1000                                    # 2.6e8 pixels: the two-range schedule applies
    movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=70, seed=31, device="cuda")
    counts = movie.view(torch.int16).to(torch.int32) & 0xffff
    box, roi, min_ng, baseline = 7, None, 5000.0, 100.0
    if case == "u16_box5_roi": box, roi = 5, (10, 10, 499, 503)            # emitters from pixel 12 on: maxima in the crop's row / column 2
    elif case == "u16_box9": box = 9
    elif case == "u16_box13": box = 13                  # (a threshold that lets many shot-noise maxima through: most waves keep deciding)
    elif case == "u8_box7": movie, min_ng, baseline = (counts // 8).clamp(max=255).to(torch.uint8), 600.0, 12.0
    elif case == "i16_box7_roi": movie, roi, baseline = (counts - 700).to(torch.int16), (9, 9, 508, 470), -600.0
    elif case == "low_threshold": min_ng = 300.0
    del counts
    torch.cuda.synchronize()
    code = backend.dtype_code({torch.uint16: np.uint16, torch.uint8: np.uint8, torch.int16: np.int16}[movie.dtype])

    def run(defer, ranges, cap_):
        _lib.check(L.pmi_localize_set_defer(defer), "pmi_localize_set_defer")
        _lib.check(L.pmi_localize_set_ranges(ranges), "pmi_localize_set_ranges")
        table = torch.zeros((_lib.PMI_LOC_COLUMNS, cap_), dtype=torch.int32, device="cuda")
        d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
        r = (ctypes.c_int64 * 4)(*roi) if roi else None
        rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), code, F, 512, 512, box, min_ng, r, 0, F - 1, baseline, 1.0, 1.0,
                                    1e-3, 100, _lib.MLE_METHODS["sigmaxy"], ctypes.c_void_p(table.data_ptr()), cap_,
                                    ctypes.c_void_p(d_n.data_ptr()), None)
        _lib.check(rc, "pmi_localize_mle_dev")
        torch.cuda.synchronize()
        return table, int(d_n.item())

    try:
        cap = 400 * F if case == "low_threshold" else 160 * F
        ref, n_ref = run(0, 1, cap)
        assert n_ref > 5000, n_ref
        if case == "tight_cap":
            cap = n_ref                      # exactly the rows: the candidates live in the library's scratch
        for ranges in (1, 2):
            got, n_got = run(1, ranges, cap)
            assert n_got == n_ref, (case, ranges, n_got, n_ref)
            assert torch.equal(got[:, :n_got], ref[:, :n_ref]), (case, ranges)
    finally:
        _lib.check(L.pmi_localize_set_ranges(2), "pmi_localize_set_ranges")
        _lib.check(L.pmi_localize_set_defer(1), "pmi_localize_set_defer")
