"""TOOL: which float32-loop failures escape the flag rules?"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from study import fuzz_spots, traces


def analyse(box, method, eps, max_it, n, rng, style="fuzz"):
    spots = fuzz_spots(box, n, rng, style)
    tr, ir, tf, aux, itf = traces(spots, eps, max_it, method)
    NP = 6 if method == "sigmaxy" else 5
    tested = [0, 1, 4, 5] if NP == 6 else [0, 1]
    T = tf.shape[1]
    # D per iteration of the fast loop
    Df = np.abs(np.diff(tf[:, :, tested], axis=1)).max(axis=2)       # (n, T-1): D of iteration k+1
    Dr = np.abs(np.diff(tr[:, :, tested], axis=1)).max(axis=2)
    kidx = np.arange(1, T)[None, :]
    valid_f = kidx <= itf[:, None]
    same = ir == itf
    fr = tr[np.arange(n), np.minimum(ir, T - 1)]; ff = tf[np.arange(n), np.minimum(itf, T - 1)]
    conv = same & (ir < max_it) & np.all(np.isfinite(fr), axis=1)
    dd = np.abs(fr - ff)[:, [0, 1, 4, 5]].max(axis=1)
    with np.errstate(invalid="ignore"):
        dph = np.abs(fr[:, 2] - ff[:, 2]) / np.maximum(np.abs(fr[:, 2]), 1)
    tol = max(1e-3, eps)
    fail = (~same) | (conv & ((dd > tol) | (dph > 1e-2)))
    # current rules
    ulp = 2.0 ** (np.floor(np.log2(max(1.0, box / 2.0))) - 23)
    margin = max(eps * 1e-3, 4 * ulp)
    wide = np.maximum(1.0, (kidx - 1) * 0.0625)
    with np.errstate(invalid="ignore"):
        r_margin = (valid_f & (np.abs(Df - eps) < margin * wide)).any(axis=1)
        den = aux[:, :T - 1, 6:6 + NP]
        r_curv = (valid_f[:, :, None] & (den >= 0)).any(axis=(1, 2))
        r_slow = (itf >= 32) | ((itf >= max_it) & ~(Df[np.arange(n), np.minimum(itf, T - 1) - 1] < eps))
        sig = tf[:, 1:, 4:4 + (2 if NP == 6 else 1)]
        r_narrow = (valid_f[:, :, None] & (sig < 0.3)).any(axis=(1, 2))
    flagged = r_margin | r_curv | r_slow | r_narrow
    return dict(spots=spots, tr=tr, ir=ir, tf=tf, aux=aux, itf=itf, Df=Df, Dr=Dr, fail=fail, flagged=flagged, same=same, dd=dd,
                rules=dict(margin=r_margin, curv=r_curv, slow=r_slow, narrow=r_narrow), valid_f=valid_f)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    rng = np.random.default_rng(11)
    for style in ("fuzz", "real"):
        for box in (7, 13, 15, 21, 5, 3):
            for method in ("sigmaxy", "sigma"):
                for eps, max_it in ((1e-3, 100), (1e-2, 100), (1e-3, 5), (1e-4, 100), (1e-4, 1000), (1e-2, 5)):
                    if max_it == 1000 and box > 7: continue
                    a = analyse(box, method, eps, max_it, n, rng, style)
                    esc = a["fail"] & ~a["flagged"]
                    print(f"{style} {box} {method} eps {eps} max_it {max_it}: fail {int(a['fail'].sum())} flagged {a['flagged'].mean():.3f} "
                          f"({', '.join(f'{k} {v.mean():.3f}' for k, v in a['rules'].items())}) ESCAPED {int(esc.sum())}", flush=True)
                    for r in np.flatnonzero(esc)[:3]:
                        print("   row", r, "it", a["itf"][r], a["ir"][r], "dd", a["dd"][r], "th_f", np.round(a["tf"][r, min(a["itf"][r], a["tf"].shape[1]-1)], 4),
                              "th_r", np.round(a["tr"][r, min(a["ir"][r], a["tr"].shape[1]-1)], 4))
