import sys
import numpy as np
sys.path.insert(0, 'tools/emul')
from rules2 import evaluate
n = int(sys.argv[1]); style = sys.argv[2]; seed = int(sys.argv[3])
rng = np.random.default_rng(seed)
tot = 0
for box in (21, 15, 7, 13, 3, 5):
    for method in ("sigma", "sigmaxy"):
        for eps, max_it in ((1e-3, 100), (1e-2, 100), (1e-3, 5), (1e-4, 100)):
            tot += int(evaluate(box, method, eps, max_it, n, rng, style)["esc"].sum())
print("TOTAL ESCAPED", tot)
