#!/usr/bin/env python3
"""Re-run the spots tools/fuzz_parity.py dumped (gpurun_out/fuzz_fail/mle_*.npz): default mode, strict mode, oracle; which
flags the float32 loop raised.  usage: python tools/check_fail_dumps.py [dir]"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from picasso_amd import backend as be  # noqa: E402

d = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/mle_fuzz_regressions"
for path in sorted(glob.glob(os.path.join(d, "mle_*.npz"))):
    z = np.load(path)
    spots, method, eps, max_it = z["spots"], str(z["method"]), float(z["eps"]), int(z["max_it"])
    o = orc.gaussmle(spots, eps, max_it, method, threads=2)
    g = be.gaussmle_arrays(spots, eps, max_it, method)
    refit, why = be.last_refit_count(), be.last_flag_reasons()
    be.set_mle_mode("strict")
    s = be.gaussmle_arrays(spots, eps, max_it, method)
    be.set_mle_mode("refit")
    for r in range(len(spots)):
        print(os.path.basename(path), "box", spots.shape[1], method, eps, max_it, "row", r, "it default/strict/oracle", int(g[3][r]), int(s[3][r]), int(o[3][r]),
              "strict==oracle bits", bool(np.array_equal(s[0][r].view(np.uint32), o[0][r].view(np.uint32))), "refit", refit, why,
              "\n    default", g[0][r].tolist(), "\n    strict ", s[0][r].tolist(), "\n    oracle ", o[0][r].tolist(), flush=True)
