"""Times pmi_peak_fit (the bounded Gaussian fit of RCC's correlation windows, one wavefront per pair) on n windows.
usage: python tools/time_peak_fit.py [n] [box]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from picasso_amd import backend as be  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
box = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(1)
h = box // 2
y, x = np.mgrid[-h:h + 1, -h:h + 1]
rois = np.empty((n, box, box))
for t in range(n):
    a = rng.uniform(5, 500); xc, yc = rng.uniform(-0.7, 0.7, 2); s = rng.uniform(0.6, 2.5); b = rng.uniform(0, 50)
    rois[t] = np.abs(a * np.exp(-0.5 * ((x - xc) ** 2 + (y - yc) ** 2) / s ** 2) + b + rng.normal(0, 0.02 * a, (box, box)))
for rep in range(3):
    t0 = time.perf_counter()
    popt, status = be.peak_fit_arrays(rois)
    dt = time.perf_counter() - t0
    print(f"n={n} box={box}: {dt * 1e3:.2f} ms per call (host windows in, parameters out), statuses {np.bincount(status.clip(0), minlength=5)}")
