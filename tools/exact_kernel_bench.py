"""KoopmanKernelRegressor.fit (the exact-kernel comparator of benchmark_lqr_hjb.py:334-381) at N samples: intermediates in
HBM (default) against the host-composed version (NYSKOOP_EXACT_HOST=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (N, 1)); u = rng.uniform(-1, 1, (N, 1))
y = x + 0.01 * (-x ** 3 + u)
X = np.hstack([x, u])
for rep in range(2):
    reg = nk.KoopmanKernelRegressor(1, kernel=nk.KernelWrapper([0.5]), gamma=1e-8)
    t0 = time.perf_counter(); reg.fit(X, y); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); P = reg.predict(X[:64]); dp = time.perf_counter() - t1
    print(f"N={N} exact-kernel fit {dt:.2f} s, predict(64) {dp * 1e3:.1f} ms, one-step error {np.abs(P - y[:64]).max():.2e} "
          f"(NYSKOOP_EXACT_HOST={os.environ.get('NYSKOOP_EXACT_HOST', '0')})", flush=True)
