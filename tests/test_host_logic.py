"""CPU tier: host-side logic of the Picasso-shaped surface (no device compute)."""
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import golden
from picasso_amd import backend, gaussmle, localize


def test_locs_from_fits_matches_reference_table():
    """Columns, dtypes, formulas and order of picasso/gaussmle.py:957-1037, against the
    table the reference produced for the bundled movie."""
    g = golden("locs_from_fits_mle")
    fit = golden("gaussmle_testdata_real")
    ids = golden("get_spots_testdata")
    idf = pd.DataFrame({"frame": ids["frame"], "x": ids["x"], "y": ids["y"], "net_gradient": ids["ng"]})
    locs = gaussmle.locs_from_fits(idf, fit["sigmaxy_theta"], fit["sigmaxy_crlb"], fit["sigmaxy_loglik"],
                                   fit["sigmaxy_iterations"], 7)
    assert list(locs.columns) == list(g["columns"])
    assert [str(locs[c].dtype) for c in locs.columns] == list(g["dtypes"])
    assert [c for c, _ in backend.LOC_COLUMNS] == list(g["columns"])
    for c in locs.columns:
        assert np.array_equal(locs[c].to_numpy(), g[c], equal_nan=True), c


def test_locs_from_fits_n_id_sort():
    idf = pd.DataFrame({"frame": [3, 1, 2], "x": [5, 6, 7], "y": [8, 9, 10],
                        "net_gradient": np.float32([1, 2, 3]), "n_id": [2, 0, 1]})
    th = np.ones((3, 6), np.float32); cr = np.ones((3, 6), np.float32)
    locs = gaussmle.locs_from_fits(idf, th, cr, np.zeros(3, np.float32), np.ones(3, np.int32), 7)
    assert list(locs["n_id"]) == [0, 1, 2] and locs["n_id"].dtype == np.uint32
    assert list(locs["frame"]) == [1, 2, 3]


def test_frame_range_and_roi_normalisation():
    assert backend.frame_range(None, 100) == (0, 100)
    assert backend.frame_range((10, 50), 100) == (10, 50)
    assert backend.frame_range((None, 50), 100) == (0, 50)
    assert backend.frame_range((5, None), 100) == (5, 100)
    assert backend.frame_range((-3, 1000), 100) == (0, 100)
    assert backend.normalise_roi(None, 32, 32) is None
    assert list(backend.normalise_roi(((2, 1), (31, 32)), 32, 32)) == [2, 1, 31, 32]
    assert list(backend.normalise_roi(((2, 1), (100, 100)), 32, 40)) == [2, 1, 32, 40]
    assert list(backend.normalise_roi(((-4, -3), (-1, -1)), 32, 40)) == [28, 37, 31, 39]


def test_unsupported_dtype_and_shape():
    with pytest.raises(TypeError, match="unsupported movie dtype"):
        backend.dtype_code(np.float64)
    with pytest.raises(ValueError):
        backend.as_movie_array(np.zeros((4, 4), np.uint16))
    big = np.zeros((2, 8, 8), dtype=">u2")
    assert backend.as_movie_array(big).dtype.isnative


def test_method_validation_happens_before_any_device_work():
    with pytest.raises(ValueError, match="Method not available"):
        gaussmle.gaussmle(np.zeros((1, 7, 7), np.float32), 1e-3, 10, method="bogus")
    with pytest.raises(ValueError, match="Method not available"):
        gaussmle.gaussmle_async(np.zeros((1, 7, 7), np.float32), 1e-3, 10, method="bogus")


class _Movie:
    """AbstractPicassoMovie-like wrapper (reference tests/conftest.py:259-319)."""
    def __init__(self, a): self._a = a; self.dtype = a.dtype
    def __len__(self): return len(self._a)
    def __getitem__(self, i): return self._a[i]
    def __iter__(self): return iter(self._a)


def _fit2d_args(movie):
    ids = pd.DataFrame({"frame": [0], "x": [10], "y": [10], "net_gradient": np.float32([1])})
    return dict(movie=movie, movie_info=[{}], camera_info={"Baseline": 0, "Sensitivity": 1, "Gain": 1, "Pixelsize": 130},
                identifications=ids, box=7)


def test_fit2d_input_assertions(testdata_movie):
    """Messages of picasso/localize.py:1416-1445 (reference tests/test_localize.py:907-952)."""
    mov = _Movie(testdata_movie)
    a = _fit2d_args(mov)
    with pytest.raises(AssertionError, match="fitting_method must be one of"):
        localize.fit2D(**a, fitting_method="bogus")
    with pytest.raises(AssertionError, match="eps must be a positive number"):
        localize.fit2D(**a, fitting_method="gaussmle", eps=-1.0)
    with pytest.raises(AssertionError, match="max_it must be a positive integer"):
        localize.fit2D(**a, fitting_method="gaussmle", max_it=0)
    with pytest.raises(AssertionError, match="mle_method"):
        localize.fit2D(**a, fitting_method="gaussmle", mle_method="x")
    with pytest.raises(AssertionError, match="movie must be a movie loaded"):
        localize.fit2D(**{**a, "movie": np.asarray(testdata_movie)}, fitting_method="gaussmle")
    with pytest.raises(AssertionError, match="movie_info must be a list"):
        localize.fit2D(**{**a, "movie_info": {}}, fitting_method="gaussmle")
    with pytest.raises(AssertionError, match="box must be a positive integer"):
        localize.fit2D(**{**a, "box": 7.0}, fitting_method="gaussmle")


def test_fit2d_missing_pixelsize_warns_then_fails_loudly_without_gpu(testdata_movie):
    from picasso_amd import _lib
    a = _fit2d_args(_Movie(testdata_movie))
    a["camera_info"] = {"Baseline": 0, "Sensitivity": 1, "Gain": 1}
    with pytest.warns(UserWarning, match="Pixelsize"):
        try:
            localize.fit2D(**a, fitting_method="gaussmle", multiprocess=False)
        except _lib.HipBackendError:
            assert _lib.device_count() == 0      # no GPU here: loud failure, no CPU fallback
    assert a["camera_info"]["Pixelsize"] == 130


def test_methods_without_a_kernel_raise_not_implemented(testdata_movie):
    """'gausslq-gpu' is the reference's CUDA Gpufit binding: refused, never silently rerouted."""
    a = _fit2d_args(_Movie(testdata_movie))
    with pytest.raises(NotImplementedError, match="Gpufit"):
        localize.fit2D(**a, fitting_method="gausslq-gpu")


def test_gausslq_fails_loudly_without_gpu(testdata_movie):
    from picasso_amd import _lib, gausslq
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipBackendError):
        gausslq.fit_spots(np.ones((2, 7, 7), np.float32))


def test_identify_deprecation_and_abort(testdata_movie):
    with pytest.warns(DeprecationWarning, match="return_info"):
        try:
            localize.identify(testdata_movie, 5000, 7, abort_callback=lambda: True)
        except Exception:
            pass
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert localize.identify(testdata_movie, 5000, 7, abort_callback=lambda: True, return_info=True) is None


def test_sigma_uncertainty_formula():
    s, so, n, bg = np.float64(1.1), np.float64(1.2), np.float64(5000.0), np.float64(20.0)
    sa2 = s**2 + 1 / 12
    tau = 2 * np.pi * sa2 * bg / n
    want = np.sqrt((s**2 / (4 * n)) * (1 + 8 * tau + np.sqrt(8 * tau / (1 + 2 * tau))))
    assert np.isclose(gaussmle.sigma_uncertainty(s, so, n, bg), want, rtol=1e-12)


def test_gausslq_table_and_precision_match_reference():
    """locs_from_fits / localization_precision of picasso/gausslq.py:404-484, 547-589 against
    the table the reference produced (em False and True)."""
    from picasso_amd import gausslq
    ids = golden("get_spots_testdata")
    idf = pd.DataFrame({"frame": ids["frame"], "x": ids["x"], "y": ids["y"], "net_gradient": ids["ng"]})
    theta = golden("gausslq_testdata_real")["theta"]
    for em in (False, True):
        g = golden("locs_from_fits_lq_em%d" % int(em))
        locs = gausslq.locs_from_fits(idf, theta, 7, em)
        assert list(locs.columns) == list(g["columns"])
        for c in locs.columns:
            assert str(locs[c].dtype) == str(g[c].dtype), c
            assert np.array_equal(locs[c].to_numpy(), g[c], equal_nan=True), c


def test_avg_table_matches_reference():
    from picasso_amd import avgroi
    ids = golden("get_spots_testdata")
    idf = pd.DataFrame({"frame": ids["frame"], "x": ids["x"], "y": ids["y"], "net_gradient": ids["ng"]})
    g = golden("avg_testdata")
    locs = avgroi.locs_from_fits(idf, g["theta"], 7, False)
    assert list(locs.columns) == list(g["columns"])
    for c in locs.columns:
        assert np.array_equal(locs[c].to_numpy(), g[c], equal_nan=True), c


def test_ensure_sanity_and_filter():
    from picasso_amd import lib, zfit
    locs = pd.DataFrame({"x": np.float32([1, 40, 3, 4, -1]), "y": np.float32([1, 2, 3, np.inf, 2]),
                         "lpx": np.float32([.1, .1, np.nan, .1, .1]), "photons": np.float32([5, 5, 5, 5, 5]),
                         "d_zcalib": np.float32([.1, .1, .1, .1, 5])})
    info = [{"Width": 32, "Height": 32, "Frames": 10}]
    out = lib.ensure_sanity(locs, info)
    assert list(out.index) == [0]
    with pytest.raises(KeyError, match="Width"):
        lib.ensure_sanity(locs, [{"Height": 3, "Frames": 2}])
    assert lib.get_from_metadata([{"a": 1}, {"a": 2}], "a") == 2
    assert lib.get_from_metadata({"a": 1}, "b", 7) == 7
    assert len(zfit.filter_z_fits(locs, 0)) == 5 and len(zfit.filter_z_fits(locs, 1)) == 4


def test_zfit_argument_checks():
    from picasso_amd import zfit
    locs = pd.DataFrame({"x": np.float32([1.0]), "sx": np.float32([1.0]), "sy": np.float32([1.0])})
    with pytest.raises(AssertionError, match="Invalid fitting method"):
        zfit.zfit(locs, [{}], calibration={}, fitting_method="x")
    with pytest.raises(AssertionError, match="Magnification factor is missing"):
        zfit.zfit(locs, [{"Pixelsize": 130}], calibration={"X Coefficients": [0] * 7})
    with pytest.raises(AssertionError, match="pixel size"):
        zfit.zfit(locs, [{}], calibration={"Magnification factor": 0.8})


def test_install_rebinds_the_reference_functions():
    """localize.install() on stand-in module objects."""
    import types
    from picasso_amd import gausslq, gaussmle as amd_mle
    pl, pm, pq = types.SimpleNamespace(), types.SimpleNamespace(), types.SimpleNamespace()
    localize.install(pl, pm, pq)
    assert pl.identify is localize.identify and pl.get_spots is localize.get_spots
    assert pl._fit2d_gaussmle is localize._fit2d_gaussmle and pl._fit2d_gausslq is localize._fit2d_gausslq
    assert pm.gaussmle is amd_mle.gaussmle and pm.gaussmle_async is amd_mle.gaussmle_async
    assert pq.fit_spots is gausslq.fit_spots and pq.fit_spots_parallel is gausslq.fit_spots_parallel
    # the rows next to the path: z fit, render (rotated views stay with the reference), RCC undrift
    from picasso_amd import imageprocess, postprocess, render, zfit
    calls = []
    pz, pi, pp = types.SimpleNamespace(), types.SimpleNamespace(), types.SimpleNamespace()
    pr = types.SimpleNamespace(_render_gaussian=lambda *a, **k: calls.append(("theirs", a, k)) or (0, None),
                               _render_hist=lambda *a, **k: calls.append(("theirs_hist", a, k)) or (0, None))
    localize.install(pl, pm, pq, pz, pr, pi, pp)
    assert pz._fit_z is zfit._fit_z and pi.rcc is imageprocess.rcc and pp.undrift is postprocess.undrift
    assert pr._render_gaussian("L", 1, 0, 0, 8, 8, 0.0, ang=(0.1, 0, 0))[0] == 0 and calls[-1][0] == "theirs"
    assert pr._render_gaussian("L", 1, 0, 0, 8, 8, 0.0, (0.1, 0, 0))[0] == 0 and len(calls) == 2
    assert pr._render_hist("L", 1, 0, 0, 8, 8, (0.1, 0, 0))[0] == 0 and calls[-1][0] == "theirs_hist"
    with pytest.raises(Exception):          # unrotated: ours, which needs the device (and a DataFrame)
        pr._render_gaussian("L", 1, 0, 0, 8, 8, 0.0)
    assert len(calls) == 3


def test_localize_streamed_chunking_logic(monkeypatch, testdata_movie):
    """Host logic of the chunked upload (no GPU): the device calls are replaced by the oracle's identify, so what
    is checked is the chunk boundaries, the frame labels, frame bounds, the progress callback, the reuse of one
    staging allocations and the empty result; the device side of the same function is the GPU tier's."""
    from oracle import oracle as orc

    class FakeStage:
        made = 0

        def __init__(self, chunk):
            FakeStage.made += 1
            self.capacity = np.asarray(chunk).nbytes
            self.load(chunk)

        def load(self, chunk):
            chunk = np.ascontiguousarray(chunk)
            assert chunk.nbytes <= self.capacity
            self.ptr, self.dtype, self.shape = chunk, chunk.dtype, chunk.shape

        def free(self):
            self.ptr = None

    def fake_device(stage_ptr, dtype, shape, box, min_ng, camera, *a, roi=None, **k):
        assert stage_ptr.shape == tuple(shape)
        fr, y, x, ng = orc.identify(stage_ptr, min_ng, box, roi=roi, threads=2)
        cols = {name: np.zeros(len(fr), dt) for name, dt in backend.LOC_COLUMNS}
        cols["frame"] = fr.astype(np.uint32)
        cols["x"], cols["y"], cols["net_gradient"] = x.astype(np.float32), y.astype(np.float32), ng
        return cols

    class Dummy:
        handle = None
        def destroy(self): pass
        def free(self): pass

    from picasso_amd import _lib
    bound = []
    monkeypatch.setattr(backend, "DeviceMovie", FakeStage)
    monkeypatch.setattr(backend, "DeviceStream", Dummy)
    monkeypatch.setattr(backend, "DeviceWorkspace", Dummy)
    monkeypatch.setattr(backend, "localize_mle_device", fake_device)
    monkeypatch.setattr(_lib, "current_key", lambda: (0, 0))            # devices=None: the caller's device and bank ...
    monkeypatch.setattr(_lib, "bind_thread", lambda dev, bank=0: bound.append((dev, bank)))   # ... bound in the lane's threads
    mov = np.ascontiguousarray(testdata_movie)
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    params = {"Min. Net Gradient": 3000, "Box Size": 7}
    whole = pd.DataFrame(fake_device(mov, mov.dtype, mov.shape, 7, 3000, cam))
    seen = []
    got = localize.localize_streamed(mov, cam, params, chunk_bytes=13 * mov[0].nbytes, progress_callback=seen.append)
    assert FakeStage.made == 2 and seen == [13, 26, 39, 52, 65, 78, 91, 100]      # two staging allocations, reused
    assert len(got) == len(whole) > 20 and got.equals(whole)

    class FrameOnly:
        def __init__(self, a): self.a = a
        def __len__(self): return len(self.a)
        def __getitem__(self, i):
            if not isinstance(i, (int, np.integer)):
                raise TypeError
            return self.a[i]

    got = localize.localize_streamed(FrameOnly(mov), cam, params, chunk_bytes=40 * mov[0].nbytes, frame_bounds=(7, 61))
    ref = whole[(whole.frame >= 7) & (whole.frame <= 61)].reset_index(drop=True)
    assert got.equals(ref) and len(ref) > 5
    none = localize.localize_streamed(mov, cam, params, frame_bounds=(50, 10))
    assert len(none) == 0 and list(none.columns) == [n for n, _ in backend.LOC_COLUMNS]
    with pytest.raises(ValueError):
        localize.localize_streamed(mov, cam, params, fitting_method="avg")


def test_streamed_scheduler_deals_chunks_over_lanes_in_frame_order():
    """`localize._run_lanes`, the scheduler behind `localize_streamed(devices=[...])`, with fake lanes (no GPU): chunks are
    dealt round robin, every lane works in its own host thread with a bound worker, a lane reuses two staging slots and
    never loads a slot its worker still reads, the tables come back in frame order whatever order the lanes finish in, the
    progress callback counts frames monotonically, an abort stops new chunks and returns None, and an error of one lane
    reaches the caller after the other lanes have closed.  Reference for the contract: the worker threads of
    picasso/localize.py:424-454 + the sort at :478, the abort at :462-470."""
    import threading
    import time

    class FakeLane:
        def __init__(self, name, delay=0.0, fail_at=None):
            self.name, self.delay, self.fail_at = name, delay, fail_at
            self.bound, self.uploads, self.ran = set(), [], []
            self.busy = [False, False]
            self.opened = self.closed = 0
            self.slots = [None, None]

        def bind(self):
            self.bound.add(threading.current_thread())      # (the Thread object: the OS may hand a finished thread's ident to the next)

        def open(self):
            assert threading.current_thread() in self.bound
            self.opened += 1

        def upload(self, k, chunk):
            assert threading.current_thread() in self.bound and not self.busy[k], "a staging slot was loaded while its chunk ran"
            self.slots[k] = np.array(chunk)
            self.uploads.append((k, int(chunk[0, 0, 0])))

        def run(self, k, c0):
            assert threading.current_thread() in self.bound
            self.busy[k] = True
            time.sleep(self.delay)
            frames = self.slots[k][:, 0, 0].astype(np.uint32)
            if self.fail_at is not None and c0 >= self.fail_at:
                self.busy[k] = False
                raise RuntimeError(f"lane {self.name} failed at {c0}")
            self.ran.append(c0)
            self.busy[k] = False
            assert int(frames[0]) == c0
            return pd.DataFrame({"frame": frames, "lane": np.full(len(frames), self.name)})

        def close(self):
            self.closed += 1

    F = 103
    movie = np.arange(F, dtype=np.uint16)[:, None, None] * np.ones((1, 2, 2), np.uint16)      # pixel value = frame number
    chunks = [(c0, min(F, c0 + 10)) for c0 in range(0, F, 10)]
    # three lanes of very different speed
    lanes = [FakeLane(0, 0.02), FakeLane(1, 0.0), FakeLane(2, 0.005)]
    seen = []
    parts = localize._run_lanes(movie, chunks, lanes, progress_callback=seen.append)
    got = pd.concat(parts, ignore_index=True)
    assert got["frame"].tolist() == list(range(F))
    assert got["lane"].tolist() == [(f // 10) % 3 for f in range(F)]
    assert seen == sorted(seen) and seen[-1] == F and len(seen) == len(chunks)
    for li, lane in enumerate(lanes):
        assert lane.opened == lane.closed == 1 and len(lane.bound) == 2          # its thread and its worker
        assert lane.ran == [c0 for i, (c0, _) in enumerate(chunks) if i % 3 == li]
        assert [k for k, _ in lane.uploads] == [n & 1 for n in range(len(lane.uploads))]
    assert len({t for lane in lanes for t in lane.bound}) == 6                 # no thread shared between lanes
    # one lane: a lane thread + one worker as well — binding is sticky, so never the calling thread
    solo = FakeLane(9)
    parts = localize._run_lanes(movie, chunks, [solo])
    assert pd.concat(parts, ignore_index=True)["frame"].tolist() == list(range(F))
    assert threading.current_thread() not in solo.bound and len(solo.bound) == 2
    # abort after the fourth chunk has been handed out: None, every lane closed
    lanes = [FakeLane(0, 0.001), FakeLane(1, 0.001)]
    asked = [0]

    def abort():
        asked[0] += 1
        return asked[0] > 4
    assert localize._run_lanes(movie, chunks, lanes, abort_callback=abort) is None
    assert all(lane.closed == 1 for lane in lanes) and sum(len(lane.ran) for lane in lanes) <= 5
    # an error in one lane: raised in the caller, the other lane stopped and closed
    lanes = [FakeLane(0, 0.001), FakeLane(1, 0.001, fail_at=30)]
    with pytest.raises(RuntimeError, match="lane 1 failed"):
        localize._run_lanes(movie, chunks, lanes)
    assert all(lane.closed == 1 for lane in lanes)


def test_install_fused_and_default_devices(monkeypatch):
    """install(fused=True, devices=...): the reference module's `localize` becomes this package's (one PCIe crossing),
    and the device list becomes the default of localize_streamed; PICASSO_AMD_DEVICES does the same for a process that
    cannot pass arguments; an explicit `devices=` wins."""
    import types
    pl, pm = types.SimpleNamespace(localize="theirs"), types.SimpleNamespace()
    monkeypatch.setattr(localize, "_default_devices", None)
    localize.install(pl, pm, types.SimpleNamespace())
    assert pl.localize == "theirs" and localize._resolve_devices(None) is None
    localize.install(pl, pm, types.SimpleNamespace(), fused=True, devices=[1, 0])
    assert pl.localize is localize.localize and localize._resolve_devices(None) == [1, 0]
    assert localize._resolve_devices([3]) == [3]
    localize.set_devices(None)
    monkeypatch.setenv("PICASSO_AMD_DEVICES", "0, 2,3")
    assert localize._resolve_devices(None) == [0, 2, 3]
    monkeypatch.setenv("PICASSO_AMD_DEVICES", "all")
    assert localize._resolve_devices(None) == "all"
    monkeypatch.delenv("PICASSO_AMD_DEVICES")
    assert localize._resolve_devices(None) is None


def test_streamed_lanes_for_devices(monkeypatch):
    """`devices=` of localize_streamed -> (device, scratch bank) lanes: None names the calling thread's own device and bank
    (what it bound itself to, else the thread's current device, bank 0) so that the lane's threads bind to it, a
    device named twice gets the two banks the library has per device, a third time / an absent device is refused."""
    from picasso_amd import _lib
    monkeypatch.setattr(_lib, "device_count", lambda: 4)
    monkeypatch.setattr(_lib, "current_key", lambda: (2, 1))
    assert localize._lanes_for(None, 7) == [(2, 1)]
    assert localize._lanes_for("all", 7) == [(0, 0), (1, 0), (2, 0), (3, 0)]
    assert localize._lanes_for([2, 0, 2], 7) == [(2, 0), (0, 0), (2, 1)]
    assert localize._lanes_for([0, 1, 2, 3], 2) == [(0, 0), (1, 0)]             # no more lanes than chunks
    for bad in ([0, 0, 0], [4], [-1], [], "gpu"):
        with pytest.raises(ValueError):
            localize._lanes_for(bad, 7)


def test_lock_is_keyed_by_device_and_bank(monkeypatch):
    """One lock per (device, scratch bank) whether the thread bound itself or not: an unbound thread on device 0 and a
    lane bound to (0, 0) write the same scratch buffers and must serialise; (0, 1) and (1, 0) run beside them."""
    import threading
    from picasso_amd import _lib
    monkeypatch.setattr(_lib, "_tls", threading.local())
    monkeypatch.setattr(_lib, "_bound_locks", {})

    class L:
        @staticmethod
        def pmi_get_device(ref):
            ref._obj.value = 0
            return 0
    monkeypatch.setattr(_lib, "load", lambda: L)
    unbound = _lib.lock()
    assert _lib.current_key() == (0, 0)
    got = {}

    def bound(key):
        _lib._tls.key = key
        got[key] = _lib.lock()
    for key in ((0, 0), (0, 1), (1, 0)):
        t = threading.Thread(target=bound, args=(key,)); t.start(); t.join()
    assert got[(0, 0)] is unbound and got[(0, 1)] is not unbound and got[(1, 0)] is not unbound
    assert _lib.bound_to() is None                      # the lane threads' binding did not leak into this thread


def test_run_lanes_leaves_the_calling_thread_unbound():
    """A single lane runs on a thread of its own too: bind() is sticky per thread, and the caller's device / bank / lock
    key must be what they were."""
    import threading
    main = threading.get_ident()
    seen = []

    class Lane:
        def bind(self): seen.append(threading.get_ident())
        def open(self): pass
        def upload(self, k, chunk): pass
        def run(self, k, c0): return pd.DataFrame({"frame": np.array([c0], np.uint32)})
        def close(self): pass
    parts = localize._run_lanes(np.zeros((6, 4, 4), np.uint16), [(0, 3), (3, 6)], [Lane()])
    assert [int(p["frame"][0]) for p in parts] == [0, 3]
    assert seen and main not in seen


def test_picks_and_locs_to_identifications_match_reference():
    """Host-side table constructors (picasso/localize.py:752-913) against the reference's output
    (tests/golden/make_goldens_surface.py); rows of one frame are compared as a set (the reference's
    sort is not stable)."""
    g = golden("surface_cases")
    picks = [tuple(p) for p in g["picks"]]
    drift = pd.DataFrame({"x": g["drift_x"], "y": g["drift_y"]})
    cols = ["frame", "x", "y", "net_gradient", "n_id"]

    def same(df, tag):
        assert list(df.columns) == cols and all(df[c].dtype == np.float64 for c in cols)
        a = np.stack([df[c].to_numpy() for c in cols], 1)
        b = np.stack([g[f"{tag}_{c}"] for c in cols], 1)
        assert a.shape == b.shape and np.all(np.diff(a[:, 0]) >= 0)
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])

    same(localize.picks_to_identifications(picks, n_frames=9), "picks_plain")
    same(localize.picks_to_identifications(picks, drift=drift), "picks_drift")
    same(localize.picks_to_identifications(picks, n_frames=17, drift=drift), "picks_drift")
    with pytest.raises(ValueError):
        localize.picks_to_identifications(picks)
    with pytest.raises(AssertionError):
        localize.picks_to_identifications(picks, n_frames=5, drift=drift)
    with pytest.raises(AssertionError):
        localize.picks_to_identifications([(1, 2, 3)], n_frames=5)
    locs = pd.DataFrame({c: g[f"l2i_in_{c}"] for c in ("frame", "x", "y")})
    same(localize.locs_to_identifications(locs, [{"Frames": 60}], 4), "l2i")
    with pytest.raises(AssertionError):
        localize.locs_to_identifications(locs, [{"Frames": 60}], -1)


def test_older_locs_from_fits_and_futures_collation():
    g = golden("surface_cases")
    ids = pd.DataFrame({c: g[f"lff_ids_{c}"] for c in ("frame", "x", "y", "net_gradient")})
    t = localize.locs_from_fits(ids, g["lff_theta"], g["lff_crlb"], g["lff_ll"], g["lff_it"], 7)
    assert list(t.columns) == list(g["lff_columns"]) and t["iterations"].dtype == np.int32 and t["frame"].dtype == np.uint32
    order = np.argsort(g["lff_frame"], kind="stable")          # input frames were sorted: the table is in input order
    for c in t.columns:
        assert t[c].dtype == g[f"lff_{c}"].dtype
        assert np.array_equal(t[c].to_numpy(), g[f"lff_{c}"][order]), c
    parts = [ids.iloc[20:], ids.iloc[:20]]
    got = localize.identifications_from_futures([localize.gausslq._DoneFuture(parts[:1]), localize.gausslq._DoneFuture(parts[1:])])
    assert np.all(np.diff(got["frame"].to_numpy()) >= 0) and len(got) == len(ids)
    assert sorted(map(tuple, got.to_numpy().tolist())) == sorted(map(tuple, ids.to_numpy().tolist()))


def test_bench_refuses_tuning_variables():
    """bench.py must not produce a line under PMI_* / PICASSO_AMD_LIB overrides unless told to (and then echoes them):
    the library's tuning variables select other kernels or another build."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PMI_MLE_MODE="fast")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "tuning variables are set" in (r.stderr + r.stdout) and "PMI_MLE_MODE" in (r.stderr + r.stdout)
    assert r.stdout.strip() == ""          # no JSON line


def test_lq_chain_scratch_layouts_are_free_of_bank_conflicts():
    """The LDS layout of the strict least-squares Jacobian's chain scratch (csrc/lq_jacobian_w.inc: LqwBox<W> and
    lqw_coff<W>) through the bank model of tools/emul/lds_chain_layout.py (lane groups and bank functions of gfx950's
    ds_read_b128 / ds_write_b64; the model reproduced SQ_LDS_BANK_CONFLICT of the 7x7 kernel before and after its
    layout changed): no chain read conflicts for any box, no write conflicts up to 9x9 — and the table the tool carries is
    the one the kernel is built with."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_chain_layout", os.path.join(root, "tools", "emul", "lds_chain_layout.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    src = open(os.path.join(root, "picasso_amd", "csrc", "lq_jacobian_w.inc")).read()
    boxes = {}
    for w, gs, h, s_expr, g_expr in re.findall(r"LqwBox<(\d+)>\s*\{ static constexpr int GS = (\d+),\s*H = (\d+), S = ([^,]+),\s*G = ([^;]+); \};", src):
        boxes[int(w)] = (int(gs), int(h), int(eval(s_expr)), int(eval(g_expr)))
    assert sorted(boxes) == [3, 5, 7, 9, 11, 13, 15, 17, 19, 21]
    assert boxes == sim.BOX
    # lqw_coff<7>: 50 c + (0, 0, 2, 6, 10, 12)
    assert "c < 2 ? 0 : (c == 2 ? 2 : (c == 3 ? 6 : (c == 4 ? 10 : 12)))" in src and sim.COFF7 == [50 * c + e for c, e in enumerate((0, 0, 2, 6, 10, 12))]
    for w, (gs, h, s, g) in boxes.items():
        rd_extra, wr_extra = sim.score(w, gs, h, s, g)
        assert rd_extra == 0, (w, rd_extra)
        if w <= 9:
            assert wr_extra == 0, (w, wr_extra)
        assert 6 * s <= g or w == 7          # the columns of a spot fit its share (7x7: its own offsets, 262 + 50 = 312)
