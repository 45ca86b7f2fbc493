"""Per-call latency distribution of small API calls (looking for sporadic stalls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
def dist(name, f, reps=300):
    f(); f()
    ts = []
    for i in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    a = np.array(ts)
    top = np.argsort(a)[-4:][::-1]
    print(f"{name}: median {np.median(a):.3f} ms, mean {a.mean():.3f}, p99 {np.percentile(a, 99):.3f}, max {a.max():.3f}; "
          f"slowest calls {[(int(i), round(float(a[i]), 2)) for i in top]}", flush=True)
for (m, d, p) in ((100, 192, 6), (500, 192, 6), (500, 6, 6)):
    A = rng.standard_normal((m, m)) * (0.9 / np.sqrt(m)); B = rng.standard_normal((m, p)); Cm = rng.standard_normal((d, m))
    for batch in (1, 16):
        z0 = rng.standard_normal((batch, m)); U = rng.standard_normal((batch, 100, p))
        dist(f"linear_rollout m={m} d={d} batch={batch}", lambda: nk.linear_rollout(A, B, Cm, z0, U))
n, d, p, m = 6000, 192, 6, 500
S = rng.standard_normal((n, d)); Uc = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + Uc @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, Uc])
reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
np.random.seed(1); reg.fit(X, Y)
x0 = X[0, :d]; Useq = rng.standard_normal((p, 100))
phi0 = reg.lift(x0.reshape(-1, 1))
Ub = np.stack([Useq.T] * 64); xb = X[:64, :d]
phib = reg.lift(xb.T).T
os.environ["NYSKOOP_TRACE"] = "1"
dist("model rollout m=500 single", lambda: reg.rollout(x0, Useq), 60)
dist("model rollout m=500 batch 64", lambda: reg.rollout(xb, Ub), 30)
dist("model rollout m=500 single again", lambda: reg.rollout(x0, Useq), 60)
