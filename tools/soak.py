"""Stability soak (GPU box): repeated full-size fits and a threaded CV sweep; watches free HBM and results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness
from bench import make_c4, CV_GRID
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n, m, d, p = 100000, 2000, 384, 6
X, Y, idx = make_c4(n, d, p, m)
dev = torch.device("cuda", 0)
Xd, Yd = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
Z = np.ascontiguousarray(Y[idx])
ref = None
free0 = None
t0 = time.perf_counter()
for i in range(reps):
    ls, gamma = CV_GRID[i % len(CV_GRID)]
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=gamma, m=m)
    reg.nystrom_centers_output = Z.T
    reg.fit(Xd, Yd)
    if i % len(CV_GRID) == 0:
        s = float(reg.A[::97, ::89].sum())
        if ref is None: ref = s
        assert s == ref, (i, s, ref)  # bitwise reproducible across repetitions
    if i == 20: free0 = torch.cuda.mem_get_info()[0]
    if i % 50 == 0: print(i, f"{time.perf_counter() - t0:.1f}s free HBM {torch.cuda.mem_get_info()[0] / 2**30:.2f} GiB", flush=True)
free1 = torch.cuda.mem_get_info()[0]
print("fits/s over the soak:", reps / (time.perf_counter() - t0), "HBM drift MiB:", (free0 - free1) / 2**20)
assert abs(free0 - free1) < 64 * 2**20
rng = np.random.default_rng(0)
S = rng.standard_normal((1010, 192)); U = rng.standard_normal((1010, 6))
Yc = np.tanh(S @ (rng.standard_normal((192, 192)) * 0.9 / np.sqrt(192))) + U @ (rng.standard_normal((6, 192)) * 0.1)
Xc = np.hstack([S, U])
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, 192), gamma=g, m=500) for l in (10., 20., 40.) for g in (1e-5, 1e-4, 1e-3)]
base = None
for rnd in range(6):
    np.random.seed(0)
    res = harness.grid_search_cv(Xc, Yc, 6, cands, n_splits=5, workers=8)
    if base is None: base = res["split_scores"]
    assert np.array_equal(base, res["split_scores"])
print("threaded CV sweeps reproducible; free HBM", torch.cuda.mem_get_info()[0] / 2**30)
