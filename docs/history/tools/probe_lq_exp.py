#!/usr/bin/env python3
"""CPU-only probe: the spots on which the device's strict least-squares mode (every sum in MINPACK's order) still differs
from the oracle — are they decided by the last bit of one float64 exp?  Runs the oracle on the saved inputs with libm's exp
and with a correctly rounded exp (orc_lq_set_exp) and compares both with the theta the device produced.
usage: python tools/probe_lq_exp.py <npz of tools/fuzz_parity.py> ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

for path in sys.argv[1:]:
    z = np.load(path)
    spots, gpu = z["spots"], z["theta_gpu"]
    out = {}
    for which in (0, 1):
        orc.lq_set_exp(which)
        out[which] = orc.gausslq(spots, full=True)
    orc.lq_set_exp(0)
    for r in range(len(spots)):
        same0 = bool(np.array_equal(out[0][0][r], gpu[r])); same1 = bool(np.array_equal(out[1][0][r], gpu[r]))
        d01 = float(np.abs(out[0][0][r][[0, 1, 4, 5]] - out[1][0][r][[0, 1, 4, 5]]).max())
        print(os.path.basename(path), "row", r, "box", int(z["box"]), "| device == oracle(libm exp):", same0, "| device == oracle(correctly rounded exp):", same1,
              "| the two oracles apart by", d01, "px, nfev", int(out[0][2][r]), int(out[1][2][r]))
