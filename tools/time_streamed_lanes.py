"""localize_streamed of config 2's movie from host memory (PCIe-inclusive): one lane against two lanes on the two scratch banks
of the same device, a few chunk sizes.  usage: python tools/time_streamed_lanes.py [frames]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from picasso_amd import localize, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mov = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda").cpu().numpy()
gb = mov.nbytes / 1e9
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
MIBS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128, 256]
params = {"Min. Net Gradient": 5000.0, "Box Size": 7}
localize.localize_streamed(mov[:500], cam, params)
for devices in (None, [0, 0], None, [0, 0]):
    for mib in (MIBS):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            locs = localize.localize_streamed(mov, cam, params, chunk_bytes=mib << 20, devices=devices)
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"devices={devices} chunk {mib} MiB: {1e3 * t:.1f} ms ({gb / t:.1f} GB/s) -> {len(locs) / t / 1e6:.2f} M loc/s", flush=True)
