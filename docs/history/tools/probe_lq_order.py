#!/usr/bin/env python3
"""CPU probe: which spots' least-squares fit depends on the ORDER of lmdif's long sums?  Runs the oracle with MINPACK's
order and with the reversed order (orc_lq_set_sum_order) on the spots of tools/diag_lq.py.
usage: python tools/probe_lq_order.py [n] [boxes]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
boxes = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [7, 3, 9, 13]


def lq_spots(box, n, seed):
    rng = np.random.default_rng(seed)
    c = box // 2
    idx = np.arange(box) - c
    x0 = rng.uniform(-1.2, 1.2, n); y0 = rng.uniform(-1.2, 1.2, n)
    sx = rng.uniform(0.6, 0.25 * box + 0.5, n); sy = rng.uniform(0.6, 0.25 * box + 0.5, n)
    gx = np.exp(-0.5 * ((idx[None] - x0[:, None]) / sx[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sx[:, None])
    gy = np.exp(-0.5 * ((idx[None] - y0[:, None]) / sy[:, None]) ** 2) / (np.sqrt(2 * np.pi) * sy[:, None])
    return rng.poisson(rng.uniform(100, 9000, n)[:, None, None] * gy[:, :, None] * gx[:, None, :] + rng.uniform(0.5, 60, n)[:, None, None]).astype(np.float32)


L = orc.lib()
L.orc_lq_set_sum_order.argtypes = [ctypes.c_int]
for box in boxes:
    spots = lq_spots(box, n, 40 + box)
    L.orc_lq_set_sum_order(0)
    a = orc.gausslq(spots, full=True, threads=8)
    gs = 8 if box <= 7 else (32 if box <= 15 else 64)
    for label, mode in (("reversed", 1), (f"device order (groups of {gs})", 2 | (gs << 2))):
        L.orc_lq_set_sum_order(mode)
        b = orc.gausslq(spots, full=True, threads=8)
        L.orc_lq_set_sum_order(0)
        diff = np.flatnonzero(~np.all((a[0] == b[0]) | (np.isnan(a[0]) & np.isnan(b[0])), axis=1))
        d = np.abs(a[0] - b[0])[:, [0, 1, 4, 5]].max(axis=1)
        print("box", box, label, ": order-dependent spots", len(diff), "rows", diff[:12].tolist(), "worst", float(np.nanmax(d)),
              "info differs", int((a[1] != b[1]).sum()), "nfev differs", int((a[2] != b[2]).sum()), flush=True)
