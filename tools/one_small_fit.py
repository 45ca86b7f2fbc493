"""Kernel trace target: a handful of CV units (fit on 808 rows + score on 202) at the cloth CV shape, one host thread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness
rng = np.random.default_rng(0)
n, d, p, m = 1010, 192, 6, 500
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
params = dict(kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-4, m=m)
idx = rng.choice(808, m, replace=False)
for _ in range(2): harness.cv_unit_score(X, Y, p, params, (0, 202), idx)
t0 = time.perf_counter()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(reps): harness.cv_unit_score(X, Y, p, params, (0, 202), idx)
print("ms per unit", (time.perf_counter() - t0) / reps * 1e3)
