"""Times the host-buffer entry points (what picasso_amd.localize.identify / fit2D use for a numpy / memmap movie):
PCIe-inclusive rates, to set beside the resident numbers of bench.py.  usage: python tools/time_host_path.py [frames]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from picasso_amd import backend as be, localize, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mov = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda").cpu().numpy()
gb = mov.nbytes / 1e9
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
for rep in range(3):
    t0 = time.perf_counter()
    fr, y, x, ng = be.identify_arrays(mov, 5000.0, 7)
    t1 = time.perf_counter()
    spots = be.get_spots_array(mov, fr, y, x, 7, cam["Baseline"], cam["Sensitivity"], cam["Gain"])
    t2 = time.perf_counter()
    th, cr, ll, it = be.gaussmle_arrays(spots, 1e-3, 100, "sigmaxy")
    t3 = time.perf_counter()
    print(f"{F} frames ({gb:.2f} GB), {len(fr)} spots: identify {1e3 * (t1 - t0):.0f} ms ({gb / (t1 - t0):.1f} GB/s), "
          f"get_spots {1e3 * (t2 - t1):.0f} ms, gaussmle {1e3 * (t3 - t2):.0f} ms -> {len(fr) / (t3 - t0) / 1e6:.2f} M loc/s end to end")
    for mib in ([None] if rep < 2 else [None, 128, 256, 512]):
        kw = {} if mib is None else {"chunk_bytes": mib << 20}
        t4 = time.perf_counter()
        locs = localize.localize_streamed(mov, cam, {"Min. Net Gradient": 5000.0, "Box Size": 7}, **kw)
        t5 = time.perf_counter()
        print(f"   localize_streamed (one upload, fused, chunk {mib or 'default'} MiB): {1e3 * (t5 - t4):.0f} ms "
              f"({gb / (t5 - t4):.1f} GB/s) -> {len(locs) / (t5 - t4) / 1e6:.2f} M loc/s")
