#!/usr/bin/env python3
"""Mint goldens for the rows of SURVEY.md section 8(f): render (N2) and RCC undrift (N1).

TEST INFRASTRUCTURE, build container only (needs /root/reference).  The reference's
``render.py`` and ``imageprocess.py`` are executed from where they lie through the shim
(tests/golden/_refshim.py).  ``postprocess.py`` and ``lib.py`` cannot be imported here (Qt,
h5py, sklearn ...); the handful of pure numpy/pandas functions needed from them
(n_segments, segment, undrift, _apply_drift, apply_drift, minimize_shifts) are compiled
from the reference files at run time, function by function, into a namespace that holds
the shim modules -- nothing of them is stored in this repository.

Run:  python tests/golden/make_goldens_postprocess.py
"""
import ast
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

warnings.simplefilter("ignore")
REF = _refshim.REF


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: " + ", ".join(f"{k}{getattr(v, 'shape', '')}" for k, v in arrays.items()))


def functions_from(path, names, namespace):
    """Compile the named top-level functions of a reference file into `namespace`."""
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=[ast.ImportFrom("__future__", [ast.alias("annotations")], 0)] + keep, type_ignores=[])
    exec(compile(ast.fix_missing_locations(mod), path, "exec"), namespace)


def drift_dataset(seed=7, frames=2000, size=64, sites=60, rate=0.35):
    """Binding sites blinking at random, a smooth drift of ~1 px over the acquisition."""
    rng = np.random.default_rng(seed)
    sx = rng.uniform(6, size - 6, sites)
    sy = rng.uniform(6, size - 6, sites)
    t = np.arange(frames)
    dx = 0.9 * np.sin(t / frames * 2.2) + 0.0003 * t
    dy = -0.7 * (t / frames) ** 2 + 0.2 * np.cos(t / frames * 5.0) - 0.2
    fr, xs, ys = [], [], []
    for f in range(frames):
        on = np.nonzero(rng.random(sites) < rate * 0.05)[0]
        for s in on:
            fr.append(f)
            xs.append(sx[s] + dx[f] + rng.normal(0, 0.06))
            ys.append(sy[s] + dy[f] + rng.normal(0, 0.06))
    n = len(fr)
    locs = pd.DataFrame({
        "frame": np.asarray(fr, np.uint32), "x": np.asarray(xs, np.float32), "y": np.asarray(ys, np.float32),
        "photons": rng.uniform(1000, 5000, n).astype(np.float32), "sx": np.full(n, 1.1, np.float32),
        "sy": np.full(n, 1.1, np.float32), "bg": np.full(n, 10, np.float32),
        "lpx": rng.uniform(0.02, 0.09, n).astype(np.float32), "lpy": rng.uniform(0.02, 0.09, n).astype(np.float32),
    })
    info = [{"Frames": frames, "Height": size, "Width": size, "Pixelsize": 130}]
    return locs, info, dx, dy


def main():
    ref = _refshim.load_reference()
    render, imageprocess = ref["render"], ref["imageprocess"]

    # ---------------- render (N2) ----------------
    rng = np.random.default_rng(11)
    N = 3000
    x = rng.uniform(-1, 33, N).astype(np.float32)
    y = rng.uniform(-1, 33, N).astype(np.float32)
    lpx = rng.uniform(0.005, 0.4, N).astype(np.float32)
    lpy = rng.uniform(0.005, 0.4, N).astype(np.float32)
    locs = pd.DataFrame({"x": x, "y": y, "lpx": lpx, "lpy": lpy})
    info = [{"Height": 32, "Width": 32, "Pixelsize": 130.0}]
    out = {"x": x, "y": y, "lpx": lpx, "lpy": lpy}
    cases = {"a": (5.0, 0.0, None), "b": (1.0, 1.0, None), "c": (7.3, 0.02, ((3.5, 2.25), (20.0, 30.5)))}
    for key, (osamp, mbw, vp) in cases.items():
        viewport = vp if vp is not None else ((0, 0), (32, 32))
        n_h, hist = render.render(locs, info, oversampling=osamp, viewport=vp, blur_method=None)
        n_g, gauss = render.render(locs, info, oversampling=osamp, viewport=vp, blur_method="gaussian", min_blur_width=mbw)
        # the same reference functions fed float64-widened inputs = numba's promotion
        (y_min, x_min), (y_max, x_max) = viewport
        image, ny, nx, xs, ys, in_view = render._render_setup(x.astype(np.float64), y.astype(np.float64), osamp,
                                                              y_min, x_min, y_max, x_max)
        bw = (osamp * np.maximum(lpx, mbw))[in_view]
        bh = (osamp * np.maximum(lpy, mbw))[in_view]
        assert bw.dtype == np.float32
        render._fill_gaussian(image, xs, ys, bw.astype(np.float64), bh.astype(np.float64), nx, ny)
        # gaussian_iso (render.py:1148-1216), NumPy execution and the float64-widened (numba) form
        n_i, iso_np = render.render(locs, info, oversampling=osamp, viewport=vp, blur_method="gaussian_iso", min_blur_width=mbw)
        image_iso, ny, nx, xs, ys, in_view = render._render_setup(x.astype(np.float64), y.astype(np.float64), osamp,
                                                                  y_min, x_min, y_max, x_max)
        s_iso = (bh + bw) / 2
        assert s_iso.dtype == np.float32 and n_i == n_g
        render._fill_gaussian(image_iso, xs, ys, s_iso.astype(np.float64), s_iso.astype(np.float64), nx, ny)
        out.update({f"{key}_iso_numpy": iso_np, f"{key}_iso_numba": image_iso})
        out.update({f"{key}_oversampling": np.float64(osamp), f"{key}_min_blur": np.float64(mbw),
                    f"{key}_viewport": np.asarray(viewport, np.float64), f"{key}_n": np.int64(n_g),
                    f"{key}_hist": hist, f"{key}_gauss_numpy": gauss, f"{key}_gauss_numba": image})
        assert n_h == n_g
    save("render_cases", **out)

    # ---------------- RCC undrift (N1) ----------------
    from scipy import interpolate
    from tqdm import trange
    ns = {"np": np, "pd": pd, "render": render, "imageprocess": imageprocess, "interpolate": interpolate,
          "trange": trange, "lib": sys.modules["picasso.lib"], "plt": None, "plot_drift": None,
          "Callable": None}
    functions_from(os.path.join(REF, "picasso", "lib.py"), {"minimize_shifts"}, ns)
    sys.modules["picasso.lib"].minimize_shifts = ns["minimize_shifts"]          # imageprocess.rcc calls lib.minimize_shifts
    functions_from(os.path.join(REF, "picasso", "postprocess.py"),
                   {"n_segments", "segment", "undrift", "_apply_drift", "apply_drift"}, ns)
    locs, info, dx, dy = drift_dataset()
    bounds, segments = ns["segment"](locs, info, 500, {"blur_method": "gaussian", "min_blur_width": 1}, lambda i: None)
    xc01 = imageprocess.xcorr(segments[0], segments[1])
    n_seg = len(segments)
    raw_y = np.zeros((n_seg, n_seg)); raw_x = np.zeros((n_seg, n_seg))
    for i in range(n_seg - 1):
        for j in range(i + 1, n_seg):
            raw_y[i, j], raw_x[i, j] = imageprocess.get_image_shift(segments[i], segments[j], 5, 32)
    shift_y, shift_x = imageprocess.rcc(list(segments), 32, lambda i: None)
    drift, undrifted = ns["undrift"](locs, info, 500, display=False, segmentation_callback=lambda i: None,
                                     rcc_callback=lambda i: None)
    save("undrift_rcc",
         frame=locs["frame"].to_numpy(), x=locs["x"].to_numpy(), y=locs["y"].to_numpy(),
         lpx=locs["lpx"].to_numpy(), lpy=locs["lpy"].to_numpy(), frames=np.int64(info[0]["Frames"]),
         size=np.int64(info[0]["Height"]), segmentation=np.int64(500), true_dx=dx, true_dy=dy,
         bounds=bounds, segments=segments, xcorr01=xc01, raw_shift_y=raw_y, raw_shift_x=raw_x,
         shift_y=shift_y, shift_x=shift_x, drift_x=drift["x"].to_numpy(), drift_y=drift["y"].to_numpy(),
         undrifted_x=undrifted["x"].to_numpy(), undrifted_y=undrifted["y"].to_numpy())
    print("n locs", len(locs), "segments", segments.shape, "max |drift - truth| x",
          np.abs((drift["x"] - drift["x"][0]) - (dx - dx[0])).max())


if __name__ == "__main__":
    main()
