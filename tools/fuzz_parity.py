"""Randomised differential test: GPU path vs the CPU oracle over random shapes, boxes, dtypes, ROIs,
frame bounds and thresholds (identify: bit-exact), random spots (gaussmle: every row on the oracle's iteration count and within 1e-3 px; gausslq: tolerances of
tests/test_gpu_parity.py) and random tables (render: ordered sums).  Prints every mismatch.
usage: python tools/fuzz_parity.py [seconds] [seed] [kinds: identify,mle,lq,render]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kinds = set((sys.argv[3] if len(sys.argv) > 3 else "identify,mle,lq,render").split(","))
rng = np.random.default_rng(seed)
t_end = time.time() + budget
fails = 0
fail_by = {}
counts = {"identify": 0, "mle": 0, "lq": 0, "render": 0}


def random_movie(dtype, F, Y, X):
    kind = rng.integers(0, 4)
    if kind == 0:
        mov = rng.poisson(rng.uniform(1, 60), size=(F, Y, X)).astype(np.float64) + rng.integers(0, 300)
    elif kind == 1:
        mov = rng.integers(0, rng.integers(2, 20), size=(F, Y, X)).astype(np.float64)          # many ties
    elif kind == 2:
        mov = rng.normal(500, 40, size=(F, Y, X))
    else:
        mov = np.zeros((F, Y, X)) + rng.integers(0, 5)
    for _ in range(int(rng.integers(0, max(2, F * Y * X // 800)))):
        f, y, x = rng.integers(0, F), rng.integers(0, Y), rng.integers(0, X)
        amp = rng.uniform(50, 6000)
        s = rng.uniform(0.7, 2.0)
        y0, y1, x0, x1 = max(0, y - 5), min(Y, y + 6), max(0, x - 5), min(X, x + 6)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        mov[f, y0:y1, x0:x1] += amp * np.exp(-0.5 * (((yy - y) / s) ** 2 + ((xx - x) / s) ** 2))
    if rng.random() < 0.2:
        mov[rng.integers(0, F), :, : X // 2] = 65535 if dtype == np.uint16 else mov.max()
    info = np.iinfo(dtype) if np.issubdtype(dtype, np.integer) else None
    if info is not None:
        mov = np.clip(np.rint(mov), info.min, info.max)
    return mov.astype(dtype)


while time.time() < t_end:
    which = rng.integers(0, 10)
    kind = "identify" if which < 5 else ("mle" if which < 7 else ("lq" if which < 9 else "render"))
    if kind not in kinds:
        continue
    try:
        if which < 5:
            dtype = [np.uint16, np.uint16, np.uint16, np.uint8, np.int16, np.float32, np.uint32, np.int32][rng.integers(0, 8)]
            box = int(rng.choice([3, 5, 7, 7, 9, 11, 13, 15]))
            F = int(rng.integers(1, 5))
            Y = int(rng.integers(box + 2, 200))
            X = int(rng.integers(max(16, box + 2), 700))
            if rng.random() < 0.6:
                X = X // 2 * 2
            mov = random_movie(dtype, F, Y, X)
            roi = None
            if rng.random() < 0.5:
                y0, x0 = int(rng.integers(0, Y // 2)), int(rng.integers(0, X // 2))
                roi = ((y0, x0), (int(rng.integers(y0 + 1, Y + 1)), int(rng.integers(x0 + 1, X + 1))))
            fb = None
            if rng.random() < 0.3:
                fb = (int(rng.integers(0, F)), int(rng.integers(0, F)))
            min_ng = float(rng.choice([-1e9, 0.0, 100.0, 1000.0, 5000.0, 30000.0]))
            a = be.identify_arrays(mov, min_ng, box, roi=roi, frame_bounds=fb)
            b = orc.identify(mov, min_ng, box, roi=roi, frame_bounds=fb, threads=4)
            counts["identify"] += 1
            if len(a[0]) != len(b[0]) or not all(np.array_equal(p, q) for p, q in zip(a, b)):
                fails += 1
                print("IDENTIFY MISMATCH", dtype.__name__, (F, Y, X), box, roi, fb, min_ng, len(a[0]), len(b[0]), flush=True)
        elif which < 7:
            box = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 19, 21]))
            n = 2048
            c = box // 2
            idx = np.arange(box)
            x0 = c + rng.uniform(-1.5, 1.5, n); y0 = c + rng.uniform(-1.5, 1.5, n)
            sx = rng.uniform(0.5, 0.3 * box + 0.5, n); sy = rng.uniform(0.5, 0.3 * box + 0.5, n)
            gx = np.exp(-0.5 * ((idx[None, :] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
            gy = np.exp(-0.5 * ((idx[None, :] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
            spots = rng.poisson(rng.uniform(20, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :]
                                + rng.uniform(0.05, 60, n)[:, None, None]).astype(np.float32)
            spots -= rng.choice(np.float32([0.0, 0.0, 3.0]), n)[:, None, None]      # a third with negative pixels
            method = ["sigmaxy", "sigma"][rng.integers(0, 2)]
            eps = float(rng.choice([1e-3, 1e-3, 1e-2, 1e-4, 3e-3, 1e-5]))
            max_it = int(rng.choice([100, 100, 5, 20, 1000 if box <= 9 else 300]))
            th, cr, ll, it = be.gaussmle_arrays(spots, eps, max_it, method)
            oth, ocr, oll, oit = orc.gaussmle(spots, eps, max_it, method, threads=orc.max_threads())
            counts["mle"] += 1
            same = it == oit
            # every row: the iteration count is the oracle's (borderline / chaotic fits are re-fitted on the device in the
            # reference's arithmetic), and where the oracle converged x, y, sigma agree to 1e-3 px
            fin = np.all(np.isfinite(oth), axis=1) & (oit < max_it)
            bad = 0
            if not same.all():
                bad = 1
            if fin.any() and np.max(np.abs(th[fin][:, [0, 1, 4, 5]] - oth[fin][:, [0, 1, 4, 5]])) > max(1e-3, eps):
                bad = 2      # (a coarser eps than 1e-3 leaves the converged position undetermined to eps)
            if bad:
                fails += 1
                dd = np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1)
                dd[~fin] = 0
                w = int(np.argmax(dd))
                if bad == 1:
                    w = int(np.flatnonzero(~same)[0])
                print("MLE MISMATCH", bad, box, method, eps, max_it, "rows differing", int((~same).sum()), "maxdiff", dd[w], "it", it[w], oit[w],
                      "gpu", np.round(th[w], 4), "orc", np.round(oth[w], 4), "sum", spots[w].sum(), "min", spots[w].min(), flush=True)
                rows = np.flatnonzero(~same | (fin & (np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1) > max(1e-3, eps))))
                __import__("os").makedirs("gpurun_out/fuzz_fail", exist_ok=True)
                np.savez_compressed(f"gpurun_out/fuzz_fail/mle_{seed}_{counts['mle']}.npz", spots=spots[rows], box=box, method=method, eps=eps,
                                    max_it=max_it, theta_gpu=th[rows], theta_orc=oth[rows], it_gpu=it[rows], it_orc=oit[rows])
        elif which < 9:
            box = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 21]))
            n = 2048 if box <= 13 else 512
            c = box // 2
            idx = np.arange(box) - c
            x0 = rng.uniform(-1.2, 1.2, n); y0 = rng.uniform(-1.2, 1.2, n)
            sx = rng.uniform(0.6, 0.25 * box + 0.5, n); sy = rng.uniform(0.6, 0.25 * box + 0.5, n)
            gx = np.exp(-0.5 * ((idx[None, :] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
            gy = np.exp(-0.5 * ((idx[None, :] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
            spots = rng.poisson(rng.uniform(100, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :]
                                + rng.uniform(0.5, 60, n)[:, None, None]).astype(np.float32)
            th, info, nfev = be.gausslq_arrays(spots, full_output=True)
            oth, oinfo, onfev = orc.gausslq(spots, full=True, threads=orc.max_threads())
            counts["lq"] += 1
            exact = np.all((th == oth) | (np.isnan(th) & np.isnan(oth)), axis=1)
            fin = np.all(np.isfinite(oth), axis=1)
            md = np.max(np.abs(th[fin][:, [0, 1, 4, 5]] - oth[fin][:, [0, 1, 4, 5]])) if fin.any() else 0
            counts["lq_spots"] = counts.get("lq_spots", 0) + n
            counts["lq_not_identical"] = counts.get("lq_not_identical", 0) + int((~exact).sum())
            dd = np.where(fin, np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1), 0)
            counts["lq_beyond_1e-3"] = counts.get("lq_beyond_1e-3", 0) + int((dd > 1e-3).sum())
            if (~exact).any():          # keep the inputs: regression cases (tests/golden/lq_fuzz_regressions)
                rows = np.flatnonzero(~exact)
                __import__("os").makedirs("gpurun_out/fuzz_fail", exist_ok=True)
                np.savez_compressed(f"gpurun_out/fuzz_fail/lq_{seed}_{counts['lq']}.npz", spots=spots[rows], box=box, theta_gpu=th[rows],
                                    theta_orc=oth[rows], info_gpu=info[rows], info_orc=oinfo[rows], worst_px=float(dd[rows].max()))
            # every spot inside the tolerance, MINPACK's verdict (info) on every spot; bit-identity is counted over the run
            if md > 1e-3 or (info != oinfo).any():
                fails += 1
                w = int(np.argmax(np.where(fin, np.abs(th[:, [0, 1, 4, 5]] - oth[:, [0, 1, 4, 5]]).max(axis=1), 0)))
                print("LQ MISMATCH", box, "not identical", int((~exact).sum()), "of", n, "maxdiff", md, "info differs", int((info != oinfo).sum()),
                      "row", w, "gpu", th[w].tolist(), "orc", oth[w].tolist(), "refit", be.last_lq_refit_count(), flush=True)
        else:
            N = int(rng.integers(1, 5000))
            H, W = int(rng.integers(8, 80)), int(rng.integers(8, 80))
            x = rng.uniform(-2, W + 2, N).astype(np.float32)
            y = rng.uniform(-2, H + 2, N).astype(np.float32)
            lpx = rng.uniform(0.005, 1.5, N).astype(np.float32)
            lpy = rng.uniform(0.005, 1.5, N).astype(np.float32)
            osamp = float(rng.choice([1.0, 2.5, 10.0, 13.3]))
            mbw = float(rng.choice([0.0, 0.03, 1.0]))
            vp = [(0.0, 0.0), (float(H), float(W))] if rng.random() < 0.6 else [(1.5, 2.25), (H - 1.0, W - 0.5)]
            n, img = be.render_arrays(x, y, osamp, vp[0][0], vp[0][1], vp[1][0], vp[1][1], lpx, lpy, mbw)
            on, oimg = orc.render(x, y, osamp, vp, lpx, lpy, "gaussian", mbw)
            counts["render"] += 1
            if n != on or img.shape != oimg.shape or (img != oimg).mean() > 2e-3 or np.max(np.abs(img - oimg)) > 2e-6 * max(1e-9, float(oimg.max())):
                fails += 1
                print("RENDER MISMATCH", N, (H, W), osamp, mbw, vp, n, on, (img != oimg).mean(), flush=True)
    except Exception as exc:            # report and keep going
        fails += 1
        print("EXCEPTION", which, repr(exc)[:300], flush=True)
print("done", counts, "failures", fails)
