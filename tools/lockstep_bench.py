"""CV-sweep throughput with lock-step batching: cloth CV shape n=1010 (808 train / 202 test), m=500,
d=192, p=6; 180 units; scores must be bit-identical to the unbatched sweep."""
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
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, d), gamma=g, m=m) for l in (8., 10., 14., 20., 28., 40.) for g in (1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3)]
nc = len(cands)
centers = {(c, f): np.random.RandomState(17 * c + f).choice(808, m, replace=False) for c in range(nc) for f in range(5)}
t0 = time.perf_counter(); base = harness.grid_search_cv(X, Y, p, cands, centers=centers); t1 = time.perf_counter()
print(f"unbatched, 1 host thread: {nc * 5} units in {t1 - t0:.3f} s = {nc * 5 / (t1 - t0):.0f} units/s")
for arg in sys.argv[1:] or ["8", "16", "32"]:
    B, G = (int(v) for v in (arg.split("x") + ["1"])[:2])
    harness.grid_search_cv(X, Y, p, cands, centers=centers, batch=B, batch_groups=G)  # warms every member (workspace, streams)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        res = harness.grid_search_cv(X, Y, p, cands, centers=centers, batch=B, batch_groups=G)
        best = min(best, time.perf_counter() - t0)
    same = np.array_equal(res["split_scores"], base["split_scores"])
    print(f"batch {B:3d} x {G} groups: {nc * 5} units in {best:.3f} s = {nc * 5 / best:.0f} units/s, bit-identical scores: {same}; {_lib.lockstep_pool(B).stats()}")
