"""Timing of the callers either side of the fit (GPU box): CV sweep units/s at the cloth CV shape, lift / predict /
rollout / closed-loop latencies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness
rng = np.random.default_rng(0)
def synth(n, d, p):
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    return np.hstack([S, U]), Y
nk.get_context(0)
# --- CV sweep, cloth CV shape (benchmark_lqr_cloth.py:46-57,159): n=1010, d=192, p=6, m=500
X, Y = synth(1010, 192, 6)
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, 192), gamma=g, m=500) for l in (10., 20., 40.) for g in (1e-5, 1e-4, 1e-3)]
np.random.seed(0)
harness.grid_search_cv(X, Y, 6, cands[:1], n_splits=5)  # warm-up
base = None
for workers in (1, 2, 4, 8):
    np.random.seed(0)
    harness.grid_search_cv(X, Y, 6, cands[:2], n_splits=5, workers=workers)  # per-thread warm-up
    np.random.seed(0)
    t0 = time.perf_counter(); res = harness.grid_search_cv(X, Y, 6, cands, n_splits=5, workers=workers); t1 = time.perf_counter()
    if base is None: base = res["split_scores"]
    assert np.array_equal(base, res["split_scores"]), "scores must not depend on the number of workers"
    print(f"CV sweep n=1010 m=500 d=192, workers={workers}: {len(cands) * 5} units in {t1 - t0:.3f} s = {len(cands) * 5 / (t1 - t0):.1f} units/s ({(t1 - t0) / (len(cands) * 5) * 1e3:.2f} ms per fit+score)")
# --- per-call latencies
for (n, d, p, m) in ((3030, 192, 6, 100), (20000, 384, 6, 2000)):
    X, Y = synth(n, d, p)
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
    np.random.seed(1); reg.fit(X, Y)
    def tm(f, reps=5):
        f(); t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    x0 = X[0, :d].reshape(-1, 1)
    U = rng.standard_normal((p, 100))
    K = rng.standard_normal((p, m)) * 1e-3
    phi0 = reg.lift(x0)
    print(f"n={n} m={m} d={d}: fit {tm(lambda: reg.fit(X, Y), 3):.2f} ms | lift(1) {tm(lambda: reg.lift(x0)):.3f} ms | predict(1000) {tm(lambda: reg.predict(X[:1000])):.3f} ms | "
          f"rollout(T=100) {tm(lambda: reg.rollout(x0, U)):.3f} ms | rollout(batch 64, T=100) {tm(lambda: reg.rollout(X[:64, :d], np.stack([U.T] * 64))):.3f} ms | "
          f"closed_loop(60) {tm(lambda: reg.closed_loop(K, phi0, phi0 * 0.9, 60)):.3f} ms | score(1000) {tm(lambda: reg.score_neg_rmse(X[:1000], Y[:1000])):.3f} ms")
