"""Diagnostic: the Jacobi-SVD fallback on the rank-deficient `inner` system of tests/golden/f9 (case a), forced through
nk_solve_spd (NYSKOOP_FORCE_PINV=1), with the per-sweep rotation counts (NYSKOOP_PINV_TRACE=1)."""
import os, sys
import numpy as np, scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd.regressors import KoopmanKernelRegressor
from oracle import nk_oracle as O
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "f9_rank_deficient.npz"))
X, Y, idx = g["X"], g["Y"], g["idx"]; d = 24; p = 3; m = 96
k = O.ThreeDimensionalKernel(3., 3., 3., d).kernel; Z = Y[idx]; Kj = k(Z, Z) + 1e-6 * np.eye(m)
Kin = np.vstack((k(Z, X[:, :d]), X[:, d:].T)); gn = float(sys.argv[1]) * X.shape[0] if len(sys.argv) > 1 else 1e-13 * X.shape[0]
inner = Kin @ Kin.T + gn * scipy.linalg.block_diag(Kj, np.eye(p))
R = np.random.default_rng(0).standard_normal((m + p, 2))
Xs = KoopmanKernelRegressor._solve_spd(nk.get_context(), inner, R)
Xo, rk = O.truncated_solve(inner, R, rcond=1e-10)
print("rank(1e-10)", rk, "rel err vs gap-cut solution", np.linalg.norm(Xs - Xo) / np.linalg.norm(Xo))
