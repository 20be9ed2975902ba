#!/usr/bin/env python3
"""The spots a fuzz run dumped (npz: spots, box, method, eps, max_it) again, strict and default mode, against the oracle.
usage: python docs/history/tools/check_dumped_mle.py <dir>"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import oracle as orc
from picasso_amd import backend as be

for path in sorted(glob.glob(os.path.join(sys.argv[1], "*.npz"))):
    z = np.load(path)
    spots, eps, max_it, method = z["spots"], float(z["eps"]), int(z["max_it"]), str(z["method"])
    o = orc.gaussmle(spots, eps, max_it, method, threads=1)
    out = []
    for mode in ("strict", "refit"):
        be.set_mle_mode(mode)
        g = be.gaussmle_arrays(spots, eps, max_it, method)
        same = bool(np.all((g[0] == o[0]) | (np.isnan(g[0]) & np.isnan(o[0]))))
        out.append(f"{mode}: it {int(g[3][0])} / {int(o[3][0])} theta {'same bits' if same else 'max diff %.3g' % float(np.nanmax(np.abs(g[0] - o[0])))}"
                   + (f" reasons {be.last_flag_reasons()}" if mode == "refit" else ""))
    be.set_mle_mode("refit")
    print(os.path.basename(path), f"box {int(z['box'])} {method} eps {eps} max_it {max_it} |", " | ".join(out))
