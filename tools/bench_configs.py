#!/usr/bin/env python3
"""The other single-GPU workloads of BASELINE.json, timed resident in HBM (bench.py measures configs[1]):

  config 3: same 10k x 512 x 512 movie, gausslq least-squares path + Gaussian render at oversampling 10
  config 5: 13x13 ROI astigmatic MLE fit + zfit on a simulated z-stack (50k frames, ~5e6 spots, as BASELINE.json names it)

One JSON line per config (same vocabulary as bench.py; value = localizations/s of the whole chain).
usage: python tools/bench_configs.py [--frames 10000] [--steps 5] [--cpu-seconds 8]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--frames5", type=int, default=50000, help="frames of config 5 (50 000 x 512 x 512 = 26 GB resident, ~5e6 spots)")
    ap.add_argument("--only", type=int, default=0, help="3 or 5: just that config")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--handoff", type=int, default=0, choices=(0, 1), help="A/B: 1 lets the scan's exact stage leave the box rows for the fit (pmi_localize_set_handoff)")
    ap.add_argument("--no-lq3d", action="store_true", help="config 5: skip the least-squares twin of the route")
    ap.add_argument("--defer", type=int, default=1, choices=(0, 1), help="A/B: 0 keeps identify's exact stage in the scan (pmi_localize_set_defer)")
    args = ap.parse_args()
    import torch
    from oracle import oracle as orc
    from picasso_amd import _lib, backend, synth
    L = _lib.load()
    _lib.require_gpu()
    _lib.check(L.pmi_localize_set_defer(args.defer), "pmi_localize_set_defer")
    _lib.check(L.pmi_localize_set_handoff(args.handoff), "pmi_localize_set_handoff")
    F, H, W = args.frames, 512, 512
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    threads = min(16, len(os.sched_getaffinity(0)))

    def timed(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.steps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def kernel_ms(fn):
        """HIP-event time of the scan kernel and of the fit stage inside the library, mean of 3 instrumented calls"""
        L.pmi_set_kernel_timing(1)
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        sa, sb = [], []
        for _ in range(3):
            fn(); torch.cuda.synchronize()
            L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
            sa.append(a.value); sb.append(b.value)
        L.pmi_set_kernel_timing(0)
        return float(np.mean(sa)), float(np.mean(sb))

    def roofline(movie_bytes, scan_ms, fit_ms, n_spots, fit_name, fit_bound, bytes_per_spot):
        ach = movie_bytes / (scan_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "identify_scan_u16_fast_kernel", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                "frac": ach / 8000.0, "traffic": None,
                "kernels": {"identify_scan": {"ms": scan_ms, "algorithmic_bytes": movie_bytes, "GB/s": ach},
                            fit_name: {"ms": fit_ms, "spots_per_s": n_spots / (fit_ms * 1e-3), "bound": fit_bound,
                                       "algorithmic_bytes": bytes_per_spot * n_spots,
                                       "GB/s": bytes_per_spot * n_spots / (fit_ms * 1e-3) / 1e9}}}

    # ---------------- config 3 ----------------
    if args.only in (0, 3):
        config3(args, torch, orc, L, _lib, synth, cam, threads, timed, kernel_ms, roofline)
    if args.only in (0, 5):
        config5(args, torch, orc, L, _lib, synth, cam, threads, timed, kernel_ms, roofline)


def config3(args, torch, orc, L, _lib, synth, cam, threads, timed, kernel_ms, roofline):
    F, H, W = args.frames, 512, 512
    movie = synth.simulate_movie(F, H, W, emitters_per_frame=116, device="cuda")
    torch.cuda.synchronize()
    cap = 140 * F
    table = torch.empty((_lib.PMI_LQ_COLUMNS, cap), dtype=torch.int32, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    image = torch.empty((5120, 5120), dtype=torch.float32, device="cuda")
    d_nr = torch.zeros(1, dtype=torch.int64, device="cuda")
    col = lambda c: ctypes.c_void_p(table.data_ptr() + c * cap * 4)      # noqa: E731

    def lq():
        _lib.check(L.pmi_localize_lq_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, H, W, 7, 5000.0, None, 0, F - 1,
                                         cam["Baseline"], cam["Sensitivity"], cam["Gain"], 0, ctypes.c_void_p(table.data_ptr()),
                                         cap, ctypes.c_void_p(d_n.data_ptr()), None), "lq")

    from picasso_amd import backend
    assert backend.get_lq_mode() == "strict"
    t_lq = timed(lq)                       # the default: every sum in MINPACK's order, theta / info / nfev lmdif's on every spot
    n = int(d_n.item())
    scan_ms, fit_ms = kernel_ms(lq)
    backend.set_lq_mode("refit")           # beside it: tree sums + a second fit of the flagged spots (round 3's default)
    try:
        t_lq_refit = timed(lq)
        _, fit_ms_refit = kernel_ms(lq)
        refit_spots = backend.last_lq_refit_count()
    finally:
        backend.set_lq_mode("strict")

    def rend():
        _lib.check(L.pmi_render_gaussian_dev(col(1), col(2), col(7), col(8), n, 10.0, 0.0, 0.0, float(H), float(W), 0.0, 0,
                                             ctypes.c_void_p(image.data_ptr()), 5120, 5120, ctypes.c_void_p(d_nr.data_ptr()), None), "render")

    t_r = timed(rend)
    cpu = None
    if args.cpu_seconds > 0:
        nf = min(F, 600)
        host = movie[:nf].cpu().numpy()
        t0 = time.perf_counter()
        fr, y, x, ng = orc.identify(host, 5000.0, 7, threads=threads)
        spots = orc.get_spots(host, fr, y, x, 7, cam)
        orc.gausslq(spots, threads=threads)
        dt = time.perf_counter() - t0
        xs = table[1, :n].view(torch.float32).cpu().numpy(); ys = table[2, :n].view(torch.float32).cpu().numpy()
        lx = table[7, :n].view(torch.float32).cpu().numpy(); ly = table[8, :n].view(torch.float32).cpu().numpy()
        t0 = time.perf_counter()
        orc.render(xs, ys, 10.0, [(0, 0), (H, W)], lx, ly, "gaussian", 0.0)
        dtr = time.perf_counter() - t0
        cpu = {"value": len(fr) / (dt + dtr * len(fr) / n), "unit": "localizations/s", "cores": threads, "kind": "port",
               "sample": f"first {nf} frames ({len(fr)} spots) identify+get_spots+gausslq on {threads} threads ({dt:.2f} s) + "
                         f"the full {n}-localization render on 1 thread ({dtr:.2f} s), C restatement of the reference algorithm"}
    print(json.dumps({"metric": "localizations/sec (7x7 ROI, gausslq + Gaussian render at oversampling 10)",
                      "value": n / (t_lq + t_r), "unit": "localizations/s", "n_gpus": 1, "steps": args.steps,
                      "ms_per_step": 1e3 * (t_lq + t_r), "higher_is_better": True, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"{F}-frame 512x512 uint16 movie, {n} spots, identify + fused cut + MINPACK lmdif fit + "
                                             "11-column table, then render to 5120x5120 float32", "frames": F, "box": 7},
                      "stages_ms": {"identify+gausslq+table": 1e3 * t_lq, "render_gaussian": 1e3 * t_r},
                      "gausslq": {"mode": "strict", "fit_ms": fit_ms,
                                  "refit_mode": {"identify+gausslq+table_ms": 1e3 * t_lq_refit, "fit_ms": fit_ms_refit, "refit_spots": refit_spots,
                                                 "value": n / (t_lq_refit + t_r), "strict_over_refit": t_lq / t_lq_refit}},
                      "roofline": roofline(movie.numel() * 2, scan_ms, fit_ms, n, "gausslq_fit (cut + init + rounds of Jacobian/QR and LM step)",
                                           "fp64 valu latency (MINPACK lmdif: serial divisions / square roots); 166 B/spot algorithmic", 166.0),
                      "cpu_baseline": cpu}), flush=True)
    del movie, table, image


def config5(args, torch, orc, L, _lib, synth, cam, threads, timed, kernel_ms, roofline):
    F, H, W = args.frames5, 512, 512
    cap = 140 * F
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    movie = synth.simulate_movie(F, H, W, emitters_per_frame=116, device="cuda", sigma=(1.1, 2.4), astigmatic=True,
                                 photons=(3000.0, 9000.0), seed=synth.DEFAULT_SEED + 5)
    torch.cuda.synchronize()
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "zfit_calib3d.npz"))
    cx = torch.tensor(g["cx"], dtype=torch.float64, device="cuda"); cy = torch.tensor(g["cy"], dtype=torch.float64, device="cuda")
    table = torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device="cuda")
    zout = torch.empty((2, cap), dtype=torch.float64, device="cuda")
    col = lambda c: ctypes.c_void_p(table.data_ptr() + c * cap * 4)      # noqa: E731

    def mle13():
        _lib.check(L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, H, W, 13, 5000.0, None, 0, F - 1,
                                          cam["Baseline"], cam["Sensitivity"], cam["Gain"], 1e-3, 100, 1,
                                          ctypes.c_void_p(table.data_ptr()), cap, ctypes.c_void_p(d_n.data_ptr()), None), "mle13")

    t_m = timed(mle13)
    n = int(d_n.item())
    scan_ms, fit_ms = kernel_ms(mle13)
    from picasso_amd import backend
    refit = backend.last_refit_count()

    def zf():
        _lib.check(L.pmi_zfit_dev(col(4), col(5), n, None, ctypes.c_void_p(cx.data_ptr()), ctypes.c_void_p(cy.data_ptr()),
                                  ctypes.c_void_p(zout.data_ptr()), ctypes.c_void_p(zout.data_ptr() + cap * 8), None), "zfit")

    t_z = timed(zf)
    # ---- the lq-3d twin: the reference's 3-D default is fitting_method="gausslq" (picasso/zfit.py:300,472) -> the same movie
    # through identify(box 13) + fused cut + MINPACK lmdif (strict mode) + 11-column table, then the same z fit
    lq_table = torch.empty((_lib.PMI_LQ_COLUMNS, cap), dtype=torch.int32, device="cuda")
    lcol = lambda c: ctypes.c_void_p(lq_table.data_ptr() + c * cap * 4)      # noqa: E731

    def lq13():
        _lib.check(L.pmi_localize_lq_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, H, W, 13, 5000.0, None, 0, F - 1,
                                         cam["Baseline"], cam["Sensitivity"], cam["Gain"], 0, ctypes.c_void_p(lq_table.data_ptr()),
                                         cap, ctypes.c_void_p(d_n.data_ptr()), None), "lq13")

    lq_route = None
    if not args.no_lq3d:
        assert backend.get_lq_mode() == "strict"
        t_l = timed(lq13)
        n_l = int(d_n.item())
        lq_scan_ms, lq_fit_ms = kernel_ms(lq13)

        def zf_lq():
            _lib.check(L.pmi_zfit_dev(lcol(4), lcol(5), n_l, None, ctypes.c_void_p(cx.data_ptr()), ctypes.c_void_p(cy.data_ptr()),
                                      ctypes.c_void_p(zout.data_ptr()), ctypes.c_void_p(zout.data_ptr() + cap * 8), None), "zfit")
        t_zl = timed(zf_lq)
        lq_route = {"metric": "localizations/sec (13x13 ROI astigmatic gausslq + zfit)", "value": n_l / (t_l + t_zl),
                    "ms_per_step": 1e3 * (t_l + t_zl), "localizations": n_l, "mode": "strict",
                    "stages_ms": {"identify+gausslq+table": 1e3 * t_l, "zfit": 1e3 * t_zl},
                    "kernels_ms": {"identify_scan": lq_scan_ms, "gausslq_fit": lq_fit_ms},
                    "second_pass_spots": backend.last_lq_refit_count(), "over_mle_route": (t_l + t_zl) / (t_m + t_z)}
        mle13(); torch.cuda.synchronize()          # (d_n and the statistics back to the MLE route's for the line below)
    del lq_table
    cpu = None
    if args.cpu_seconds > 0:
        nf = min(F, 400)
        host = movie[:nf].cpu().numpy()
        t0 = time.perf_counter()
        fr, y, x, ng = orc.identify(host, 5000.0, 13, threads=threads)
        spots = orc.get_spots(host, fr, y, x, 13, cam)
        th, cr, ll, it = orc.gaussmle(spots, 1e-3, 100, "sigmaxy", threads=threads)
        orc.zfit(th[:, 4], th[:, 5], g["cx"], g["cy"], threads=threads)
        dt = time.perf_counter() - t0
        cpu = {"value": len(fr) / dt, "unit": "localizations/s", "cores": threads, "kind": "port",
               "sample": f"first {nf} frames ({len(fr)} spots) identify+get_spots+gaussmle(13x13)+zfit, {threads} threads, {dt:.2f} s"}
    print(json.dumps({"metric": "localizations/sec (13x13 ROI astigmatic MLE + zfit)", "value": n / (t_m + t_z),
                      "unit": "localizations/s", "n_gpus": 1, "steps": args.steps, "ms_per_step": 1e3 * (t_m + t_z),
                      "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"{F}-frame 512x512 uint16 astigmatic movie, {n} spots, identify(box 13) + fused cut + "
                                             "MLE sigmaxy + 17-column table, then bounded-Brent zfit", "frames": F, "box": 13},
                      "stages_ms": {"identify+gaussmle+table": 1e3 * t_m, "zfit": 1e3 * t_z},
                      "mle": {"mode": backend.get_mle_mode()[0], "refit_spots": refit, "refit_reasons": backend.last_flag_reasons(),
                              # (the library defers identify's exact stage to the fit for boxes <= 7 only: sixteen lanes per candidate cost more than they save at 13x13)
                              "exact_stage_deferred_to_fit": False},
                      "roofline": roofline(movie.numel() * 2, scan_ms, fit_ms, n, "mle_fit_13x13 (g8 init/iterate/final + strict refit + crlb)",
                                           "fp32 valu issue; 406 B/spot algorithmic", 406.0),
                      "lq3d_route": lq_route,
                      "cpu_baseline": cpu}), flush=True)


if __name__ == "__main__":
    main()
