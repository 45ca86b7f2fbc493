"""Random shapes against the oracle (GPU box): python3 tools/shape_sweep.py [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
rng = np.random.default_rng(2024)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
worst = 0.0
ms = [64, 65, 127, 128, 129, 200, 333, 500, 640, 1023, 1024, 1026, 1088]
for it in range(count):
    m = int(rng.choice(ms))
    d = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 48, 96]))
    p = int(rng.choice([0, 1, 2, 6]))
    n = int(m + rng.integers(1, 3000))
    fam = rng.choice(["rbf", "matern", "rbf3"])
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) / np.sqrt(d)))
    if p: Y = Y + 0.1 * U @ rng.standard_normal((p, d))
    X = np.hstack([S, U])
    idx = rng.choice(n, m, replace=False)
    ell = float(rng.choice([0.5, 1.0, 3.0])) * np.sqrt(d)
    if fam == "matern" or d % 3:
        kern, okern = nk.KernelWrapper([ell] * d), O.KernelWrapper([ell] * d)
        fam = "matern"
    else:
        kern, okern = nk.ThreeDimensionalKernel(ell, 1.1 * ell, 0.9 * ell, d), O.ThreeDimensionalKernel(ell, 1.1 * ell, 0.9 * ell, d)
    gamma = float(rng.choice([1e-5, 1e-3]))
    reg = nk.KoopmanNystromRegressor(p, kernel=kern, gamma=gamma, m=m); reg.nystrom_centers_output = Y.T[:, idx]
    ref = O.KoopmanNystromOracle(p, kernel=okern, gamma=gamma, m=m, faithful=False); ref.nystrom_centers_output = Y.T[:, idx]
    t0 = time.perf_counter(); reg.fit(X, Y); t1 = time.perf_counter(); ref.fit(X, Y)
    q = min(n, 300)
    e_pred = np.linalg.norm(reg.predict(X[:q]) - ref.predict(X[:q])) / np.linalg.norm(ref.predict(X[:q]))
    e_lift = np.linalg.norm(reg.lift(X[:50, :d].T) - ref.lift(X[:50, :d].T)) / np.linalg.norm(ref.lift(X[:50, :d].T))
    e_w = np.linalg.norm(reg.weights - reg.C @ np.hstack([reg.A, reg.B])) / np.linalg.norm(reg.weights)
    worst = max(worst, e_pred, e_lift)
    print(f"n={n:5d} m={m:4d} d={d:3d} p={p} {fam:6s} g={gamma:.0e}: fit {1e3*(t1-t0):6.1f} ms iters {reg.fit_stats_['sqrt_iters']:2d} "
          f"pred {e_pred:.1e} lift {e_lift:.1e} W=CG {e_w:.1e}", flush=True)
    # many landmarks in 1-3 dimensions make the regularised systems ill-conditioned (cond * eps ~ 1e-6..1e-3 on the
    # predictions for ANY solver; m = 1024 Matern landmarks in 7 dimensions: 1.5e-6); the lifted states and the identity W = C [A B] do not depend on them
    assert np.isfinite(e_pred) and e_pred < (1e-2 if d <= 3 else (1e-4 if d <= 8 else 1e-6)) and e_lift < 1e-8 and e_w < 1e-11, "mismatch"
print("worst", worst)
