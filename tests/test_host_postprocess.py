"""CPU tier: the host logic of picasso_amd.render / imageprocess / postprocess (argument handling,
callbacks, crop / peak / fit bookkeeping, spline, drift application) with the three device calls
replaced by the oracle's CPU restatements.  The GPU tier runs the same checks through the C ABI."""
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import golden
from oracle import oracle as orc
from picasso_amd import backend, imageprocess, postprocess, render


@pytest.fixture()
def cpu_device_calls(monkeypatch):
    def render_arrays(x, y, oversampling, y_min, x_min, y_max, x_max, lpx=None, lpy=None, min_blur_width=0.0, iso=False):
        method = None if lpx is None else ("gaussian_iso" if iso else "gaussian")
        return orc.render(x, y, oversampling, [(y_min, x_min), (y_max, x_max)], lpx, lpy, method, min_blur_width)

    def rcc_pairs_arrays(segments, roi, box, pairs=None):
        segments = np.asarray(segments, np.float64)
        n = len(segments)
        if pairs is None:
            pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
        peak = np.zeros((len(pairs), 2), np.int32); valid = np.zeros(len(pairs), np.int32)
        rois = np.zeros((len(pairs), box, box)); crop = (0, 0)
        for p, (i, j) in enumerate(pairs):
            w = orc.peak_window(segments[i], segments[j], box, roi)
            if w is None:
                valid[p] = -1
                continue
            peak[p] = w[0], w[1]; crop = (w[2], w[3])
            if w[4] is not None:
                valid[p] = 1; rois[p] = w[4]
        return peak, valid, rois, crop

    def rcc_shifts_arrays(segments, roi, box, pairs=None):
        segments = np.asarray(segments, np.float64)
        _, Y, X = segments.shape
        peak, valid, rois, (Y_, X_) = rcc_pairs_arrays(segments, roi, box, pairs)
        shifts = np.zeros((len(valid), 2)); status = np.full(len(valid), -1, np.int32)
        for p in range(len(valid)):
            if valid[p] == 1:
                shifts[p] = orc.image_shift_from_window(rois[p], box, int(peak[p, 0]), int(peak[p, 1]), Y_, X_, Y, X)
                status[p] = 2
        return shifts, status

    monkeypatch.setattr(backend, "rcc_shifts_arrays", rcc_shifts_arrays)
    monkeypatch.setattr(backend, "render_arrays", render_arrays)
    monkeypatch.setattr(backend, "rcc_pairs_arrays", rcc_pairs_arrays)
    monkeypatch.setattr(backend, "xcorr_array", lambda a, b: orc.xcorr(a, b))


def test_render_argument_handling(cpu_device_calls):
    g = golden("render_cases")
    locs = pd.DataFrame({k: g[k] for k in ("x", "y", "lpx", "lpy")})
    info = [{"Height": 32, "Width": 32, "Pixelsize": 130.0}]
    with pytest.warns(DeprecationWarning, match="oversampling"):
        n, img = render.render(locs, info, oversampling=5.0, blur_method="gaussian")
    assert n == int(g["a_n"]) and np.array_equal(img, g["a_gauss_numba"])
    n2, img2 = render.render(locs, info, disp_px_size=26.0, blur_method="gaussian")       # 130 / 26 = 5
    assert n2 == n and np.array_equal(img, img2)
    assert np.array_equal(render.render(locs, info, disp_px_size=26.0)[1], g["a_hist"])
    assert np.array_equal(render.render(locs, info, disp_px_size=26.0, blur_method="gaussian_iso")[1], g["a_iso_numba"])
    vp = [tuple(g["c_viewport"][0]), tuple(g["c_viewport"][1])]
    n4, img4 = render.render(locs, info, disp_px_size=130.0 / 7.3, viewport=vp, blur_method="gaussian", min_blur_width=0.02)
    assert n4 == int(g["c_n"]) and img4.shape == g["c_gauss_numba"].shape
    with pytest.raises(Exception, match="blur_method not understood"):
        render.render(locs, info, disp_px_size=26.0, blur_method="nope")
    for m in ("smooth", "convolve"):
        with pytest.raises(NotImplementedError):
            render.render(locs, info, disp_px_size=26.0, blur_method=m)
    with pytest.raises(NotImplementedError):
        render.render(locs, info, disp_px_size=26.0, blur_method="gaussian", ang=(0.1, 0, 0))
    with pytest.raises(KeyError):
        render.render(locs, [{"Height": 32, "Width": 32}], disp_px_size=26.0)
    with pytest.raises(ValueError, match="Need info"):
        render._viewport(None, None)


def test_undrift_composition_matches_reference_goldens(cpu_device_calls):
    g = golden("undrift_rcc")
    locs = pd.DataFrame({"frame": g["frame"], "x": g["x"], "y": g["y"], "lpx": g["lpx"], "lpy": g["lpy"]})
    info = [{"Frames": int(g["frames"]), "Height": int(g["size"]), "Width": int(g["size"]), "Pixelsize": 130}]
    assert postprocess.n_segments(info, 500) == 4 and postprocess.n_segments(info, 300) == 7
    seen = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bounds, segments = postprocess.segment(locs, info, 500, {"blur_method": "gaussian", "min_blur_width": 1}, seen.append)
    assert seen == [0, 1, 2, 3, 4] and np.array_equal(bounds, g["bounds"]) and segments.dtype == np.float64
    assert np.max(np.abs(segments - g["segments"])) < 2e-5 * g["segments"].max()
    assert np.max(np.abs(imageprocess.xcorr(g["segments"][0], g["segments"][1]) - g["xcorr01"])) < 1e-12 * np.abs(g["xcorr01"]).max()
    ref_segments = list(g["segments"])
    for i, j in ((0, 1), (0, 3), (2, 3)):
        sy, sx = imageprocess.get_image_shift(ref_segments[i], ref_segments[j], 5, 32)
        assert abs(sy - g["raw_shift_y"][i, j]) < 1e-6 and abs(sx - g["raw_shift_x"][i, j]) < 1e-6
    calls = []
    shift_y, shift_x = imageprocess.rcc(ref_segments, 32, calls.append)
    assert calls == list(range(7))
    assert np.max(np.abs(shift_y - g["shift_y"])) < 1e-6 and np.max(np.abs(shift_x - g["shift_x"])) < 1e-6
    seg_calls, rcc_calls = [], []
    drift, und = postprocess.undrift(locs, info, 500, display=False, segmentation_callback=seg_calls.append,
                                     rcc_callback=rcc_calls.append)
    assert seg_calls == [0, 1, 2, 3, 4] and rcc_calls == list(range(7))
    assert np.max(np.abs(drift["x"].to_numpy() - g["drift_x"])) < 2e-4 and np.max(np.abs(drift["y"].to_numpy() - g["drift_y"])) < 2e-4
    assert und["x"].dtype == np.float64 and np.max(np.abs(und["x"].to_numpy() - g["undrifted_x"])) < 2e-4
    assert locs["x"].dtype == np.float32                     # the caller's table is not modified


def test_shift_conventions_and_degenerate_pairs(cpu_device_calls):
    z = np.zeros((16, 16))
    assert imageprocess.get_image_shift(z, z, 5) == (0, 0)                       # empty images (imageprocess.py:85-86)
    a = np.zeros((16, 16)); a[3, 3] = 1.0
    b = np.zeros((16, 16)); b[10, 11] = 1.0
    assert imageprocess.get_image_shift(a, b, 5) == (0, 0)                       # truncated fit window (:116-119)
    yy, xx = np.mgrid[0:48, 0:48]
    blob = lambda cy, cx: np.exp(-0.5 * (((yy - cy) / 1.5) ** 2 + ((xx - cx) / 1.5) ** 2))   # noqa: E731
    sy, sx = imageprocess.get_image_shift(blob(20, 22), blob(22.5, 19.25), 5, 32)
    assert abs(sy - 2.5) < 0.05 and abs(sx + 2.75) < 0.05                        # shift FROM A TO B, (y, x)


def test_apply_drift_contract():
    locs = pd.DataFrame({"frame": np.array([0, 1, 1, 2], np.uint32), "x": np.ones(4, np.float32), "y": np.ones(4, np.float32)})
    info = [{"Frames": 3}]
    drift = pd.DataFrame({"x": [0.0, 0.5, 1.0], "y": [0.0, -0.5, -1.0]})
    out = postprocess.apply_drift(locs.copy(), info, drift=drift)
    assert np.allclose(out["x"], [1, 0.5, 0.5, 0]) and np.allclose(out["y"], [1, 1.5, 1.5, 2]) and out["x"].dtype == np.float64
    out = postprocess.apply_drift(locs.copy(), info, drift=drift.to_numpy())
    assert np.allclose(out["x"], [1, 0.5, 0.5, 0])
    with pytest.raises(ValueError, match="shape"):
        postprocess.apply_drift(locs.copy(), info, drift=np.zeros((2, 2)))
    with pytest.raises(ValueError, match="columns"):
        postprocess.apply_drift(locs.copy(), info, drift=pd.DataFrame({"x": [0.0] * 3}))
    with pytest.raises(AssertionError):
        postprocess.apply_drift(locs.copy(), info, drift=[1, 2, 3])
