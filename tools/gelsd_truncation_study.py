"""Evidence for DESIGN.md section 3 / tests/test_gpu_configs.py: what the reference's lstsq (LAPACK gelsd, rcond = eps,
regressors.py:155,165) does on the ill-conditioned candidates of the real cloth hyper-parameter grid
(benchmark_lqr_cloth.py:46-57: n = 1010, m = 500, fold 0), and how far ANY other solver of the same two regularised systems
lands from it.  CPU only (NumPy / SciPy + the oracle's kernel functions); run:  python tools/gelsd_truncation_study.py

For each candidate: sigma_min / sigma_max of `inner`, the rank gelsd reports, the rank LAPACK's own SVD gives with the same
eps cut-off, the ratio of extreme Cholesky pivots, and the held-out score with the regularised systems solved by
  gelsd (the reference) | gelsy | Cholesky | SVD truncated at eps | SVD truncated at 100 eps.
"""
import os, sys
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nk_oracle as O

t = np.load(os.path.join(ROOT, "tests", "golden", "cloth_trajs_all.npz"))
st = t["states_e10"] / 1e10
X = np.hstack([np.vstack((st[i][:, :-1], t["inputs"][i][:, :-1])) for i in range(10)]).T
Y = np.hstack([st[i][:, 1:] for i in range(10)]).T
lo, hi = 0, 202
tr = np.r_[0:lo, hi:1010]
Xtr, Ytr = X[tr], Y[tr]
idx = np.random.RandomState(0).choice(len(tr), 500, replace=False)
eps = np.finfo(float).eps
m, p = 500, 6


def solve(Am, Bm, mode):
    if mode == "gelsd": return scipy.linalg.lstsq(Am, Bm)[0]
    if mode == "gelsy": return scipy.linalg.lstsq(Am, Bm, lapack_driver="gelsy")[0]
    if mode == "chol": return scipy.linalg.cho_solve(scipy.linalg.cho_factor(Am), Bm)
    U, s, Vt = np.linalg.svd(Am)
    keep = s > (eps if mode == "svd" else 100 * eps) * s[0]
    return (Vt[keep].T / s[keep]) @ (U[:, keep].T @ Bm)


print("ls gamma | sv ratio | rank gelsd / svd(eps) of 506 | pivot ratio | score gelsd | rel. score difference: gelsy chol svd svd100")
for ls in [(1, 1, 1), (10, 10, 10), (100, 100, 100), (1, 10, 100)]:
    for gamma in (1e-7, 1e-6, 1e-5):
        k = O.ThreeDimensionalKernel(*ls, 192).kernel
        Z = Ytr[idx]
        Kmm = k(Z, Z); Kj = Kmm + 1e-6 * np.eye(m)
        S = scipy.linalg.sqrtm(Kj).real
        Kin = np.vstack((k(Z, Xtr[:, :192]), Xtr[:, 192:].T)); Kout = k(Z, Ytr)
        gn = gamma * len(tr)
        inner = Kin @ Kin.T + gn * scipy.linalg.block_diag(Kj, np.eye(p))
        inner_rec = gn * Kj + Kout @ Kout.T
        right = scipy.linalg.block_diag(scipy.linalg.solve(S, Kmm.T, assume_a="her").T, np.eye(p))
        left = scipy.linalg.solve(S, Kout @ Kin.T, assume_a="her")
        sv = np.linalg.svd(inner, compute_uv=False)
        rk = scipy.linalg.lstsq(inner, np.eye(m + p))[2]
        L = np.linalg.cholesky(inner); dl = np.diag(L) ** 2
        Xq = X[lo:hi]
        phi = scipy.linalg.solve(S, k(Z, Xq[:, :192]), assume_a="her")
        scores = {}
        for mode in ("gelsd", "gelsy", "chol", "svd", "svd100"):
            G = left @ solve(inner, right, mode)
            Cm = (Ytr.T @ Kout.T) @ solve(inner_rec, S, mode)
            pred = ((Cm @ G) @ np.vstack((phi, Xq[:, 192:].T))).T
            scores[mode] = -np.mean(np.sqrt(np.mean((Y[lo:hi] - pred) ** 2, axis=0)))
        r = scores["gelsd"]
        print(f"{ls} {gamma:g} | {sv[-1] / sv[0]:.1e} | {rk} / {(sv > eps * sv[0]).sum()} | {dl.min() / dl.max():.1e} | {r:.6e} | "
              + " ".join(f"{abs(scores[k_] - r) / abs(r):.1e}" for k_ in ("gelsy", "chol", "svd", "svd100")))
