import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness, _lib
rng = np.random.default_rng(0)
n, d, p, m = 1010, 192, 6, 500
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, d), gamma=g, m=m) for l in (10., 20., 40.) for g in (1e-5, 1e-4, 1e-3)]
centers = {(c, f): np.random.RandomState(17 * c + f).choice(808, m, replace=False) for c in range(9) for f in range(5)}
base = harness.grid_search_cv(X, Y, p, cands, centers=centers)
for B in [int(a) for a in sys.argv[1:]] or [4, 16]:
    harness.grid_search_cv(X, Y, p, cands[:2], centers=centers, batch=B)
    t0 = time.perf_counter()
    res = harness.grid_search_cv(X, Y, p, cands, centers=centers, batch=B)
    dt = time.perf_counter() - t0
    same = np.array_equal(res["split_scores"], base["split_scores"])
    print(f"batch {B}: 45 units in {dt:.3f} s = {45 / dt:.0f} units/s, bit-identical scores: {same}, max diff {np.abs(res['split_scores'] - base['split_scores']).max():.2e}; {_lib.lockstep_pool(B).stats()}")
if os.environ.get("LS_DIAG"):
    for B in (20, 32, 32, 45):
        res = harness.grid_search_cv(X, Y, p, cands, centers=centers, batch=B)
        diff = np.abs(res["split_scores"] - base["split_scores"])
        print("B", B, "units differing:", np.argwhere(diff > 0).tolist(), "max", diff.max())
