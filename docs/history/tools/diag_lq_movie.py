import ctypes, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from picasso_amd import backend as be, synth, _lib
L = _lib.load()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=116, device="cuda")
torch.cuda.synchronize()
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
for rep in range(3):
    t0 = time.perf_counter()
    t = be.localize_lq_device(ctypes.c_void_p(movie.data_ptr()), np.uint16, tuple(movie.shape), 7, 5000.0, cam)
    torch.cuda.synchronize()
    print("pass", rep, (time.perf_counter() - t0) * 1e3, "ms incl. host copies;", len(t["frame"]), "rows; refit", be.last_lq_refit_count(), be.last_lq_tie_reasons(), flush=True)
