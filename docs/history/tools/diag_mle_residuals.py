"""Prints, for every stored MLE fuzz residual (tests/golden/mle_fuzz_regressions), today's distance between the device
fit (default and strict mode) and the oracle.  usage: python tools/diag_mle_residuals.py"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mle_fuzz_regressions", "mle_*.npz"))):
    z = np.load(path)
    eps, max_it, method = float(z["eps"]), int(z["max_it"]), str(z["method"])
    o = orc.gaussmle(z["spots"], eps, max_it, method, threads=1)
    row = [os.path.basename(path), int(z["box"]), eps, max_it]
    for mode in ("refit", "strict"):
        be.set_mle_mode(mode)
        g = be.gaussmle_arrays(z["spots"], eps, max_it, method)
        d = np.abs(g[0] - o[0])[0]
        row += [mode, int(g[3][0]), int(o[3][0]), f"dx {d[0]:.2e} dy {d[1]:.2e} dsx {d[4]:.2e} dsy {d[5]:.2e} dN/N {d[2] / o[0][0, 2]:.2e}"]
    print(*row)
be.set_mle_mode("refit")
