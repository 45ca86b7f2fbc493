#!/usr/bin/env python3
"""SHA-256 of golden-fixture fits (small d: direct kernels) and of kernel matrices -- for comparing two library builds bit by bit."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
h = lambda *a: hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in a)).hexdigest()[:16]
G = "tests/golden/"
for name, p in (("f3_duffing_matern.npz", 1), ("f4_hjb_matern.npz", 1), ("f8_hjb_config2.npz", 1), ("f1_cloth_rbf_wellcond.npz", 6), ("f2_synth_rbf_d384.npz", 6)):
    g = np.load(G + name)
    X, Y = g["X"].astype(np.float64), g["Y"].astype(np.float64)
    d = Y.shape[1]
    ls = np.atleast_1d(g["ls"]).astype(np.float64)
    if "matern" in name or "config2" in name:
        kern = nk.KernelWrapper(ls if ls.size == d else np.repeat(ls, d))
    else:
        l3 = ls if ls.size == 3 else np.repeat(ls, 3)
        kern = nk.ThreeDimensionalKernel(*l3, d)
    reg = nk.KoopmanNystromRegressor(p, kernel=kern, gamma=float(g["gamma"]), m=len(g["idx"]))
    reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, g["idx"]])
    reg.fit(X, Y)
    q = X[:64]
    print(name, h(reg.A, reg.B, reg.C), h(reg.predict(q)), h(kern.kernel(X[:300, :d], Y[:200])), flush=True)
