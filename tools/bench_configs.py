#!/usr/bin/env python3
"""The other single-GPU workloads of BASELINE.json, timed resident in HBM (bench.py measures configs[1]):

  config 3: same 10k x 512 x 512 movie, gausslq least-squares path + Gaussian render at oversampling 10
  config 5: 13x13 ROI astigmatic MLE fit + zfit on a simulated z-stack (10k frames here, ~1e6 spots)

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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()
    import torch
    from oracle import oracle as orc
    from picasso_amd import _lib, backend, synth
    L = _lib.load()
    _lib.require_gpu()
    F, H, W = args.frames, 512, 512
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    threads = min(16, len(os.sched_getaffinity(0)))

    def timed(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.steps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    # ---------------- config 3 ----------------
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

    t_lq = timed(lq)
    n = int(d_n.item())

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
                      "cpu_baseline": cpu}), flush=True)
    del movie, table, image

    # ---------------- config 5 ----------------
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

    def zf():
        _lib.check(L.pmi_zfit_dev(col(4), col(5), n, None, ctypes.c_void_p(cx.data_ptr()), ctypes.c_void_p(cy.data_ptr()),
                                  ctypes.c_void_p(zout.data_ptr()), ctypes.c_void_p(zout.data_ptr() + cap * 8), None), "zfit")

    t_z = timed(zf)
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
                      "stages_ms": {"identify+gaussmle+table": 1e3 * t_m, "zfit": 1e3 * t_z}, "cpu_baseline": cpu}), flush=True)


if __name__ == "__main__":
    main()
