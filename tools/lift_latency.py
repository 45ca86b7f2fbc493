"""Latency of lifting / predicting ONE state (the per-tick work of a controller that closes the loop on a real plant)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
def tm(f, reps=200):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
for (n, d, p, m) in ((3030, 192, 6, 100), (4000, 2, 1, 100), (6000, 192, 6, 500)):
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    X = np.hstack([S, U])
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d) if d % 3 == 0 else nk.KernelWrapper([1.0] * d), gamma=1e-5, m=m)
    np.random.seed(1); reg.fit(X, Y)
    x1 = X[:1, :d].T.copy(); xa = X[:1].copy(); x8 = X[:8, :d].T.copy()
    print(f"m={m} d={d}: lift(1) {tm(lambda: reg.lift(x1)):.1f} us | lift(8) {tm(lambda: reg.lift(x8)):.1f} us | predict(1) {tm(lambda: reg.predict(xa)):.1f} us", flush=True)
