import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rules import analyse
np.set_printoptions(linewidth=250, precision=4, suppress=False)
box, method, eps, max_it, seed = int(sys.argv[1]), sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 50000
rng = np.random.default_rng(seed)
a = analyse(box, method, eps, max_it, n, rng)
esc = np.flatnonzero(a["fail"] & ~a["flagged"])
print("escaped", esc)
for r in esc[:4]:
    K = max(a["itf"][r], a["ir"][r])
    print("row", r, "itf", a["itf"][r], "itr", a["ir"][r])
    for k in range(0, K + 1):
        d = a["tf"][r, k] - a["tr"][r, k]
        aux = a["aux"][r, k] if k < K else None
        print(k, "th_f", a["tf"][r, k], "d", d, "Df", a["Df"][r, k - 1] if k else 0, "Dr", a["Dr"][r, k - 1] if k else 0)
        if aux is not None:
            print("    num", aux[:6], "den", aux[6:12], "Q", aux[12:18])
