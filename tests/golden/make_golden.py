#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py            # needs /root/reference (read-only) on sys.path

What is imported from the reference: `regressors` (KoopmanNystromRegressor, ThreeDimensionalKernel,
KernelWrapper, LinearKernelWrapper) and `dynamical_systems` (HJB plant).  `benchmark_lqr_*.py` cannot be
imported (python-control is not installed, and must not be stubbed), so harness-level vectors come from
(a) scikit-learn's real GridSearchCV driving the reference estimator and (b) the reference's shipped
result CSVs (K_lqr_seed_*.csv), both recorded below.

Only DATA is written: inputs, landmark indices, kernel parameters and the reference's outputs.
Nothing from the reference's source text is copied.
"""
import os
import sys

import numpy as np

REF = os.environ.get("NK_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import regressors as R  # noqa: E402  (the reference)
import dynamical_systems as DS  # noqa: E402
from sklearn.model_selection import GridSearchCV  # noqa: E402


def fit_ref(kernel, X, Y, idx, gamma, m, p):
    reg = R.KoopmanNystromRegressor(p, kernel=kernel, gamma=gamma, m=m)
    reg.nystrom_centers_output = Y.T[:, idx]
    reg.fit(X, Y)
    return reg


def ref_rollout(reg, x0, controls):
    """Open-loop forecast through the reference object's own lift/A/B/C (what validate_dyn_sys iterates)."""
    z = reg.lift(x0.reshape(-1, 1))
    zs = [z]
    for i in range(controls.shape[1] - 1):
        z = reg.A @ z + reg.B @ controls[:, i].reshape(-1, 1)
        zs.append(z)
    Z = np.hstack(zs)
    return reg.C @ Z, Z


def pack_fit(reg, X, Y, idx, extra=None, nq=7, store_inputs=True, in_dtype=np.float64):
    d = Y.shape[1]
    q = np.linspace(0, X.shape[0] - 1, nq).astype(int)
    out = dict(idx=idx.astype(np.int64), A=reg.A, B=reg.B, C=reg.C, W=reg.weights, q=q,
               lift=reg.lift(X[q, :d].T), predict=reg.predict(X[q]),
               K_mn_out_head=reg.kernel.kernel(reg.nystrom_centers_output.T, Y[:16]))
    if store_inputs:
        out["X"] = X.astype(in_dtype)
        out["Y"] = Y.astype(in_dtype)
    if extra:
        out.update(extra)
    return out


def cloth_data(trajs):
    Xs, Ys, raw = [], [], []
    for i in trajs:
        tr = np.loadtxt(f"{REF}/8x8_cloth_swing_xyz/state_samples_cloth_swing_{i}.csv", delimiter=",").T
        u = np.loadtxt(f"{REF}/8x8_cloth_swing_xyz/input_samples_cloth_swing_{i}.csv", delimiter=",")[:, :6].T
        raw.append((tr, u))
        Xs.append(np.vstack((tr[:, :-1], u[:, :-1])))
        Ys.append(tr[:, 1:])
    return np.hstack(Xs).T.copy(), np.hstack(Ys).T.copy(), raw


def main():
    # ---- F1: cloth subsample, anisotropic RBF (ThreeDimensionalKernel), ill- and well-conditioned twins
    X, Y, raw = cloth_data([10, 11, 12])  # n = 303, d = 192, p = 6
    rs = np.random.RandomState(7)
    idx = rs.choice(X.shape[0], 32, replace=False)
    test_traj, test_u = cloth_data([0])[2][0]
    for tag, ls, gamma in (("illcond", (10.0, 10.0, 10.0), 1e-7), ("wellcond", (1.0, 10.0, 100.0), 1e-3)):
        kern = R.ThreeDimensionalKernel(*ls, 192)
        reg = fit_ref(kern, X, Y, idx, gamma, 32, 6)
        sim, Z = ref_rollout(reg, test_traj[:, 0], test_u)
        rmse_abs = np.sqrt(np.mean(np.square(test_traj - sim)))
        np.savez_compressed(f"{OUT}/f1_cloth_rbf_{tag}.npz", **pack_fit(
            reg, X, Y, idx, extra=dict(ls=np.array(ls), gamma=gamma, test_traj=test_traj, test_u=test_u,
                                       rollout=sim, rollout_lifted=Z, rmse_abs=rmse_abs)))

    # ---- F2: synthetic C4-shaped (d=384, p=6), isotropic RBF l=20, inputs stored as float32-exact values
    rng = np.random.default_rng(1234)
    n, d, p, m = 1024, 384, 6, 128
    S = rng.standard_normal((n, d))
    U = rng.standard_normal((n, p))
    Wt = rng.standard_normal((d, d)) * 0.9 / np.sqrt(d)
    Bt = rng.standard_normal((p, d)) * 0.1
    Y = (np.tanh(S @ Wt) + U @ Bt).astype(np.float32).astype(np.float64)
    X = np.hstack([S, U]).astype(np.float32).astype(np.float64)
    np.random.seed(0)
    idx = np.random.choice(np.arange(n), size=m, replace=False)
    kern = R.ThreeDimensionalKernel(20.0, 20.0, 20.0, d)
    reg = fit_ref(kern, X, Y, idx, 1e-6, m, p)
    Useq = rng.standard_normal((p, 50))
    sim, Z = ref_rollout(reg, X[5, :d], Useq)
    np.savez_compressed(f"{OUT}/f2_synth_rbf_d384.npz", **pack_fit(
        reg, X, Y, idx, in_dtype=np.float32,
        extra=dict(ls=np.array([20.0]), gamma=1e-6, x0=X[5, :d], Useq=Useq, rollout=sim, rollout_lifted=Z)))

    # ---- F3: Duffing CSV subsample, Matern-5/2 l=[1,1] (benchmark_lqr_classic.py:47), d=2, p=1
    xf = np.loadtxt(f"{REF}/duffing/duffing_x_forced.csv", delimiter=",")
    uf = np.loadtxt(f"{REF}/duffing/duffing_u_forced.csv", delimiter=",").reshape(1, -1)
    yf = np.loadtxt(f"{REF}/duffing/duffing_y_forced.csv", delimiter=",")
    sel = np.arange(0, 20000, 19)[:1024]
    X = np.vstack((xf, uf))[:, sel].T.copy()
    Y = yf[:, sel].T.copy()
    rs = np.random.RandomState(3)
    idx = rs.choice(1024, 50, replace=False)
    kern = R.KernelWrapper([1, 1])
    reg = fit_ref(kern, X, Y, idx, 1e-6, 50, 1)
    # one contiguous piece of a forced trajectory as rollout check (relative-% RMSE form, classic:39)
    tr = np.hstack((xf[:, :199], yf[:, 198:199]))
    uu = uf[:, :200]
    sim, Z = ref_rollout(reg, tr[:, 0], uu)
    rmse_rel = np.sqrt(np.sum(np.square(tr - sim))) / np.sqrt(np.sum(np.square(sim))) * 100
    np.savez_compressed(f"{OUT}/f3_duffing_matern.npz", **pack_fit(
        reg, X, Y, idx, extra=dict(ls=np.array([1.0, 1.0]), gamma=1e-6, test_traj=tr, test_u=uu,
                                   rollout=sim, rmse_rel=rmse_rel)))

    # ---- F4: HJB regenerated (benchmark_lqr_hjb.py:110-125,153-176 recipe, seed 0), Matern l=0.1, d=1, p=1
    plant = DS.HJB(Ts=0.01, name="hjb", n_states=1, n_inputs=1, state_lb=-1.0, state_ub=1.0,
                   input_lb=[-1], input_ub=[1])
    np.random.seed(0)
    n_trajs, n_samp = 4, 200
    Xh = np.zeros((2, n_trajs * n_samp))
    Yh = np.zeros((1, n_trajs * n_samp))
    k = 0
    for _ in range(n_trajs):
        x = np.random.uniform(plant.state_lb, plant.state_ub)
        for _ in range(n_samp):
            u = np.random.uniform(plant.input_lb, plant.input_ub).reshape(1, 1)
            Xh[:, k] = np.squeeze(np.vstack((x, u)))
            x = plant.update_SOM(x, u).reshape(-1, 1)
            Yh[:, k] = np.squeeze(x)
            k += 1
    X, Y = Xh.T.copy(), Yh.T.copy()
    rs = np.random.RandomState(11)
    idx = rs.choice(800, 40, replace=False)
    kern = R.KernelWrapper([0.1])
    reg = fit_ref(kern, X, Y, idx, 1e-4, 40, 1)
    np.savez_compressed(f"{OUT}/f4_hjb_matern.npz", **pack_fit(
        reg, X, Y, idx, extra=dict(ls=np.array([0.1]), gamma=1e-4)))

    # ---- F4c: exact kernel estimator (KoopmanKernelRegressor, regressors.py:58-111) on the first 300 HJB samples
    Xk, Yk = X[:300], Y[:300]
    kreg = R.KoopmanKernelRegressor(1, kernel=R.KernelWrapper([0.5]), gamma=1e-4)
    kreg.fit(Xk, Yk)
    qk = np.linspace(-0.9, 0.9, 7).reshape(1, -1)
    np.savez_compressed(f"{OUT}/f4c_hjb_exact_kernel.npz", X=Xk, Y=Yk, ls=np.array([0.5]), gamma=1e-4,
                        W=kreg.weights, C=kreg.C, lift=kreg.lift(qk), q=qk, predict=kreg.predict(Xk[:9]))

    # ---- F4b: linear kernel (LinearKernelWrapper, regressors.py:28-30) kernel-matrix values only
    Aq = np.random.RandomState(5).standard_normal((9, 7))
    Bq = np.random.RandomState(6).standard_normal((13, 7))
    np.savez_compressed(f"{OUT}/f4b_kernels.npz", A=Aq, B=Bq,
                        linear=R.LinearKernelWrapper(0.7).kernel(Aq, Bq),
                        matern=R.KernelWrapper(np.linspace(0.5, 2.0, 7)).kernel(Aq, Bq),
                        rbf3d=R.ThreeDimensionalKernel(0.5, 1.5, 3.0, 7).kernel(Aq, Bq),
                        rbf_self=R.ThreeDimensionalKernel(0.5, 1.5, 3.0, 7).kernel(Aq, Aq))

    # ---- F5: GridSearchCV (real sklearn) over the reference estimator, n_jobs=1 so that the global legacy
    #          RNG is consumed fit by fit: landmarks per (candidate, fold) are reproducible from the seed.
    X, Y, _ = cloth_data([0, 1, 2, 3])  # n = 404 (CV shape of benchmark_lqr_cloth.py:159, shortened)
    cands = [(1.0, 10.0, 100.0), (10.0, 10.0, 10.0)]
    gammas = [1e-5, 1e-3]
    kernels = [R.ThreeDimensionalKernel(*c, 192) for c in cands]
    clf = GridSearchCV(R.KoopmanNystromRegressor(6), {"kernel": kernels, "gamma": gammas, "m": [24]},
                       scoring="neg_root_mean_squared_error", n_jobs=1)
    np.random.seed(42)
    clf.fit(X, Y)
    res = clf.cv_results_
    order_gamma = np.array([float(g) for g in res["param_gamma"]])
    order_kernel = np.array([kernels.index(kk) for kk in res["param_kernel"]])
    split_scores = np.stack([res[f"split{k}_test_score"] for k in range(5)], axis=1)  # (n_cand, 5)
    np.savez_compressed(f"{OUT}/f5_cloth_gridsearch.npz", X=X, Y=Y, cands=np.array(cands),
                        order_gamma=order_gamma, order_kernel=order_kernel, m=24, seed=42,
                        split_scores=split_scores, mean_test_score=res["mean_test_score"],
                        best_index=clf.best_index_)

    # ---- F6: shipped known-answer gains (8x8_cloth_swing_xyz/sim_results/nystrom/data/K_lqr_seed_*.csv)
    gains = {f"K_lqr_seed_{s}": np.loadtxt(
        f"{REF}/8x8_cloth_swing_xyz/sim_results/nystrom/data/K_lqr_seed_{s}.csv") for s in (0, 1)}
    _, _, raw = cloth_data(range(10, 40))  # the training set of benchmark_lqr_cloth.py:153-156,218-220
    np.savez_compressed(f"{OUT}/f6_cloth_known_gain.npz", trajs=np.stack([r[0] for r in raw]),
                        inputs=np.stack([r[1] for r in raw]), **gains)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
