"""Per-candidate diagnostics on the real cloth CV grid (f7 inputs, fold 0): time, square-root iterations, ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
g = np.load("tests/golden/f7_cloth_cv_full.npz"); t = np.load("tests/golden/cloth_trajs_all.npz")
st = t["states_e10"] / 1e10
X = np.ascontiguousarray(np.hstack([np.vstack((st[i][:, :-1], t["inputs"][i][:, :-1])) for i in range(10)]).T)
Y = np.ascontiguousarray(np.hstack([st[i][:, 1:] for i in range(10)]).T)
idx = np.random.RandomState(0).choice(808, 500, replace=False)
slow = []
for c in range(81):
    ls = g["ls_grid"][int(g["order_kernel"][c])]; gam = float(g["order_gamma"][c])
    reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(*ls, 192), gamma=gam, m=500)
    reg.nystrom_centers_output = Y[idx + 202].T
    t0 = time.perf_counter(); reg.fit(X, Y, row_ranges=[(202, 1010)], fetch=False); dt = time.perf_counter() - t0
    s = reg.fit_stats_
    if dt > 0.006 or s["rank_inner"] < 506 or s["rank_inner_rec"] < 500:
        slow.append((c, tuple(ls), gam, round(dt * 1e3, 1), s["sqrt_iters"], s["rank_inner"], s["rank_inner_rec"]))
print(len(slow), "slow / truncated candidates of 81:")
for r in slow: print(r)
