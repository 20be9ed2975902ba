"""Iteration counts of the MLE fit on the benchmark's movie (config 2): how long are the longest fits — what bounds the
re-fit kernel's launch?  usage: python tools/diag_iterations.py [frames] [box] [eps]"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from picasso_amd import backend, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
box = int(sys.argv[2]) if len(sys.argv) > 2 else 7
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
dev = torch.device("cuda", 0)
movie = synth.simulate_movie(F, 512, 512, emitters_per_frame=100, seed=synth.DEFAULT_SEED, device=dev)
torch.cuda.synchronize()
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
out = backend.localize_mle_device(movie.data_ptr(), np.uint16, (F, 512, 512), box, 5000.0, cam, eps=eps)
it = out["iterations"]
print("spots", len(it), "re-fitted", backend.last_refit_count(), backend.last_flag_reasons())
print("mean iterations", it.mean(), "max", it.max())
for t in (10, 15, 20, 30, 40, 60, 80, 100):
    print(f"  >= {t:3d}: {int((it >= t).sum())}")
