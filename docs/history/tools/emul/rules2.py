"""TOOL: candidate flag rules for the float32 loop, evaluated on emulated traces."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from study import fuzz_spots, traces


def evaluate(box, method, eps, max_it, n, rng, style, TOP=16.0, R=0.9, slow=32, verbose=True, margin_rel=1e-3, margin_ulps=4):
    spots = fuzz_spots(box, n, rng, style)
    tr, ir, tf, aux, itf = traces(spots, eps, max_it, method)
    NP = 6 if method == "sigmaxy" else 5
    tested = [0, 1, 4, 5] if NP == 6 else [0, 1]
    T = tf.shape[1]
    st = -np.diff(tf[:, :, :NP].astype(np.float64), axis=1)            # (n, T-1, NP): step of iteration k+1 (theta_old - theta_new)
    Df = np.abs(st[:, :, tested]).max(axis=2)
    kidx = np.arange(1, T)[None, :]
    valid = kidx <= itf[:, None]
    same = ir == itf
    fr = tr[np.arange(n), np.minimum(ir, T - 1)]; ff = tf[np.arange(n), np.minimum(itf, T - 1)]
    conv = same & (ir < max_it) & np.all(np.isfinite(fr), axis=1)
    dd = np.abs(fr - ff)[:, [0, 1, 4, 5]].max(axis=1)
    with np.errstate(invalid="ignore"):
        dph = np.abs(fr[:, 2] - ff[:, 2]) / np.maximum(np.abs(fr[:, 2]), 1)
    tol = max(1e-3, eps)
    fail = (~same) | (conv & ((dd > tol) | (dph > 1e-2)))
    ulp = 2.0 ** (np.floor(np.log2(max(1.0, box / 2.0))) - 23)
    margin = max(eps * margin_rel, margin_ulps * ulp)
    wide = np.maximum(1.0, (kidx - 1) * 0.0625)
    with np.errstate(invalid="ignore"):
        r_margin = (valid & (np.abs(Df - eps) < margin * wide)).any(axis=1)
        den = aux[:, :T - 1, 6:6 + NP]
        r_curv = (valid[:, :, None] & (den >= 0)).any(axis=(1, 2))
        r_slow = itf >= slow
        sig = tf[:, 1:, 4:4 + (2 if NP == 6 else 1)]
        r_narrow = (valid[:, :, None] & (sig < 0.3)).any(axis=(1, 2))
        r_top = (valid & (aux[:, :T - 1, 30] >= TOP)).any(axis=1)
        # oscillation: sign flip with a step that did not shrink below R x the previous one, above the rounding floor
        thabs = np.abs(tf[:, 1:, :NP].astype(np.float64))
        floor = 16 * 2.0 ** -23 * np.maximum(thabs, 1e-3)
        a0, a1 = st[:, :-1], st[:, 1:]
        osc = (a0 * a1 < 0) & (np.abs(a1) > R * np.abs(a0)) & (np.abs(a1) > floor[:, 1:]) & valid[:, 1:, None]
        r_osc = osc.any(axis=(1, 2))
    rules = dict(margin=r_margin, curv=r_curv, slow=r_slow, narrow=r_narrow, top=r_top, osc=r_osc)
    flagged = np.zeros(n, bool)
    for v in rules.values(): flagged |= v
    esc = fail & ~flagged
    if verbose:
        print(f"{style} {box} {method} eps {eps} max_it {max_it}: fail {int(fail.sum())} flagged {flagged.mean():.4f} "
              f"({', '.join(f'{k} {v.mean():.4f}' for k, v in rules.items())}) ESCAPED {int(esc.sum())}", flush=True)
        for r in np.flatnonzero(esc)[:3]:
            print("   row", r, "it", itf[r], ir[r], "dd", dd[r], "th_f", np.round(ff[r], 4), "th_r", np.round(fr[r], 4))
    return dict(esc=esc, fail=fail, flagged=flagged, rules=rules, tf=tf, tr=tr, itf=itf, ir=ir, aux=aux, spots=spots)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    styles = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fuzz", "real"]
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    tot = 0
    for style in styles:
        for box in (7, 13, 15, 21, 5, 3, 9, 17):
            for method in ("sigmaxy", "sigma"):
                for eps, max_it in ((1e-3, 100), (1e-2, 100), (1e-3, 5), (1e-4, 100), (1e-2, 5), (1e-4, 1000)):
                    if max_it == 1000 and box > 7: continue
                    tot += int(evaluate(box, method, eps, max_it, n, rng, style)["esc"].sum())
    print("TOTAL ESCAPED", tot)
