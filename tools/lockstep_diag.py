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
kern = nk.ThreeDimensionalKernel(10., 10., 10., d)
idx = np.random.RandomState(0).choice(808, m, replace=False)
def unit(k):
    reg = nk.KoopmanNystromRegressor(p, kernel=kern, gamma=1e-5, m=m)
    rows = idx + 202
    reg.nystrom_centers_output = Y[rows].T
    reg.fit(X, Y, row_ranges=[(0, 0), (202, n)], fetch=False)
    sc = reg.score_neg_rmse(X[:202], Y[:202])
    st = reg.fit_stats_
    return sc, st["sqrt_iters"], st["sqrt_residual"], st["rank_inner"], float(np.abs(reg.A).sum()), float(np.abs(reg.C).sum())
base = unit(0)
print("unbatched", base)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pool = _lib.lockstep_pool(B)
for rep in range(3):
    out = pool.map(unit, range(B))
    bad = [(k, o) for k, o in enumerate(out) if o != base]
    print("rep", rep, "bad members:", len(bad), bad[:4])
