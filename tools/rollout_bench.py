"""Latency of the callers' loops on the device: rollout / closed loop, single and batched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
stalls = []
def tm(f, reps=20):
    """median over `reps` calls (a sporadic slow call is listed separately, not averaged in)"""
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(ts))
    stalls.extend((round(t, 2), i) for i, t in enumerate(ts) if t > 5 * med)
    return med
for (n, d, p, m) in ((3030, 192, 6, 100), (3030, 192, 6, 128), (4000, 2, 1, 100), (6000, 192, 6, 500)):
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    X = np.hstack([S, U])
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d) if d % 3 == 0 else nk.KernelWrapper([1.0] * d), gamma=1e-5, m=m)
    np.random.seed(1); reg.fit(X, Y)
    x0 = X[0, :d]; Useq = rng.standard_normal((p, 100)); K = rng.standard_normal((p, m)) * 1e-3
    phi0 = reg.lift(x0.reshape(-1, 1))
    Ub = np.stack([Useq.T] * 64); xb = X[:64, :d]
    phib = reg.lift(xb.T).T
    t1 = tm(lambda: reg.rollout(x0, Useq)); t64 = tm(lambda: reg.rollout(xb, Ub), 5)
    c1 = tm(lambda: reg.closed_loop(K, phi0, phi0 * 0.9, 60)); c64 = tm(lambda: reg.closed_loop(K, phib, phib * 0.9, 60), 5)
    print(f"m={m} d={d} p={p}: rollout T=100 single {t1:.3f} ms | batch 64 {t64:.3f} ms ({t64 / 64:.4f} per trajectory, 64 singles = {64 * t1:.2f} ms) | "
          f"closed loop 60 steps single {c1:.3f} ms | batch 64 {c64:.3f} ms | lift(1) {tm(lambda: reg.lift(x0.reshape(-1, 1))):.3f} ms")
    if stalls: print(f"  calls slower than 5 x the median (ms, index): {stalls}"); stalls.clear()
