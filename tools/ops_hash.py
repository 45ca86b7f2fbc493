#!/usr/bin/env python3
"""SHA-256 of the operators of a few fits and of two kernel-block products (library given by NYSKOOP_LIB): two builds that
compute the same bits print the same lines."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
h = lambda *a: hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in a)).hexdigest()[:16]
rng = np.random.default_rng(5)
for (n, d, p, m, fam) in ((30000, 384, 6, 1100, "rbf"), (25000, 64, 2, 1030, "matern"), (3000, 40, 3, 700, "rbf")):
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    X = np.hstack([S, U])
    kern = nk.ThreeDimensionalKernel(6., 7., 8., d) if fam == "rbf" and d % 3 == 0 else nk.KernelWrapper([6.0] * d)
    reg = nk.KoopmanNystromRegressor(p, kernel=kern, gamma=1e-5, m=m)
    reg.nystrom_centers_output = np.ascontiguousarray(Y[:m].T)
    reg.fit(X, Y)
    print(n, d, p, m, fam, h(reg.A, reg.B, reg.C), "finite", bool(np.all(np.isfinite(reg.A))), flush=True)
