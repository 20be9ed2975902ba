import sys
import numpy as np
sys.path.insert(0, 'tools/emul')
from study import fuzz_spots, traces
u = 2.0 ** -24
rng = np.random.default_rng(21)
for style in ("fuzz", "real"):
    for box in (7, 13, 21, 5):
        for method in ("sigmaxy", "sigma"):
            n = 40000
            spots = fuzz_spots(box, n, rng, style)
            tr, ir, tf, aux, itf = traces(spots, 1e-3, 3, method, T=4)
            NP = 6 if method == "sigmaxy" else 5
            e = np.abs(tf[:, 1, :NP].astype(np.float64) - tr[:, 1, :NP])
            num, den, Q, An, Ad, top = aux[:, 0, :NP].astype(np.float64), aux[:, 0, 6:6 + NP].astype(np.float64), aux[:, 0, 12:12 + NP], aux[:, 0, 18:18 + NP].astype(np.float64), aux[:, 0, 24:24 + NP].astype(np.float64), aux[:, 0, 30]
            delta = num / den
            est = u * (An + np.abs(delta) * Ad) / np.abs(den) + u * np.abs(tf[:, 1, :NP])     # + half ulp of the stored theta
            with np.errstate(invalid="ignore", divide="ignore"):
                ratio = e / est
            ok = np.isfinite(ratio)
            out = []
            for l in range(NP):
                r = ratio[:, l][ok[:, l]]
                out.append(f"{np.percentile(r, 99):.1f}/{np.percentile(r, 99.99):.1f}/{r.max():.1f}")
            print(style, box, method, "ratio true/est p99/p99.99/max per param:", " ".join(out), " top p50/p99/max", np.percentile(top, 50), np.percentile(top, 99), top.max(), flush=True)
