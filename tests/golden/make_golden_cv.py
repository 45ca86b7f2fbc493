#!/usr/bin/env python3
"""Golden vectors for the hyper-parameter searches of configs 1 and 2 (learn_hyperparams of benchmark_lqr_classic.py:44-64 and
benchmark_lqr_hjb.py:47-70), produced by RUNNING THE REFERENCE in the build container: scikit-learn's real GridSearchCV
(n_jobs = 1, so that the landmark draws from the global legacy RNG come in a reproducible order) over the reference estimator.

  f13  Duffing: 20 validation trajectories x 200 steps (classic:205-207: n = 20 x int(2 // 0.01) = 3980, d = 2, p = 1), Matern-5/2 l = [1, 1],
       16 gammas 1e-6 .. 10^-2.25, m = 500, 5 folds = 80 units, sklearn's 'neg_root_mean_squared_error'
  f14  HJB: 20 x 200 steps (hjb:205-207: n = 3980, d = 1, p = 1), 3 Matern kernels l = 0.01 / 0.1 / 1 x 16 gammas x 5 folds = 240 units

plus, unit by unit, the reference's own reproducibility as in make_golden_envelope.py: `spread` (inputs perturbed by one part in
1e15) and `envelope` (the same reference code with gelsy / Cholesky / eigen-solve for its two solves).  Data protocol of this
fixture (the authors' validation sets are drawn after 200 test trajectories from a state of the RNG that is not worth
replaying: their search results are only shipped as a pickle): np.random.seed(0) before generate_dataset, np.random.seed(1)
before the search.

    python tests/golden/make_golden_cv.py [duffing] [hjb]
"""
import os
import sys
import time

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden_configs as MG  # noqa: E402  (imports the reference)
import make_golden_duffing as MD  # noqa: E402
from make_golden_envelope import LstsqSwap  # noqa: E402

R, DS = MG.R, MG.DS
from sklearn.model_selection import GridSearchCV  # noqa: E402

GAMMAS = np.power(10.0, np.arange(-6, -2, 0.25))


def duffing_dataset(ds, n_trajs, n_samp):
    """benchmark_lqr_classic.py:100-119 (generate_dataset) without the plotting."""
    X = np.zeros((ds.n_states + 1, n_trajs * n_samp))
    Y = np.zeros((ds.n_states, n_trajs * n_samp))
    k = 0
    for _ in range(n_trajs):
        length = np.sqrt(np.random.uniform(0, ds.radius_sampling))
        angle = np.pi * np.random.uniform(0, ds.angle_sampling)
        x = np.array([length * np.cos(angle), length * np.sin(angle)]).reshape([-1, 1])
        for _ in range(n_samp):
            u = np.random.uniform(ds.input_lb, ds.input_ub).reshape([ds.n_inputs, 1])
            X[:, k] = np.squeeze(np.vstack((x, u)))
            x = ds.update_SOM(x, u)
            Y[:, k] = np.squeeze(x)
            k += 1
    return X.T.copy(), Y.T.copy()


def search(X, Y, n_inputs, kernels, m):
    """the split scores of GridSearchCV in its candidate order, with the (kernel index, gamma) of every candidate"""
    np.random.seed(1)
    clf = GridSearchCV(R.KoopmanNystromRegressor(n_inputs), {"kernel": kernels, "gamma": GAMMAS, "m": [m]},
                       scoring="neg_root_mean_squared_error", n_jobs=1)
    clf.fit(X, Y)
    res = clf.cv_results_
    sc = np.stack([res[f"split{f}_test_score"] for f in range(5)], axis=1)
    ok = np.array([kernels.index(p["kernel"]) for p in res["params"]])
    og = np.array([p["gamma"] for p in res["params"]])
    return sc, ok, og, res["mean_test_score"]


def with_bars(name, X, Y, n_inputs, make_kernels, m, extra):
    t0 = time.time()
    sc, ok, og, mean = search(X, Y, n_inputs, make_kernels(), m)
    print(f"{name}: {sc.size} units in {time.time() - t0:.0f} s; best candidate {int(np.argmax(mean))}", flush=True)
    rel = lambda a: np.abs(a - sc) / np.abs(sc)
    rng = np.random.default_rng(7)
    sp, _, _, _ = search(X * (1 + 1e-15 * rng.standard_normal(X.shape)), Y * (1 + 1e-15 * rng.standard_normal(Y.shape)),
                         n_inputs, make_kernels(), m)
    spread = rel(sp)
    env = np.zeros_like(sc)
    for mode in ("gelsy", "chol", "eigh"):
        with LstsqSwap(mode):
            se, _, _, _ = search(X, Y, n_inputs, make_kernels(), m)
        env = np.maximum(env, rel(se))
        print(f"{name}: {mode} moves the scores by up to {rel(se).max():.2e} (median {np.median(rel(se)):.2e})", flush=True)
    print(f"{name}: 1e-15 perturbation moves them by up to {spread.max():.2e} (median {np.median(spread):.2e})")
    np.savez_compressed(f"{OUT}/{name}.npz", X=X, Y=Y, m=m, seed=1, split_scores=sc, order_kernel=ok, order_gamma=og,
                        mean_test_score=mean, spread=spread, envelope=env, **extra)


def duffing():
    ds = MD.duffing_plant()
    np.random.seed(0)
    X, Y = duffing_dataset(ds, 20, int(2 // ds.Ts))
    with_bars("f13_duffing_cv", X, Y, 1, lambda: [R.KernelWrapper([1, 1])], 500, dict(ls_grid=np.array([[1.0, 1.0]])))


def hjb():
    plant = DS.HJB(Ts=0.01, name="hjb", n_states=1, n_inputs=1, state_lb=-1.0, state_ub=1.0, input_lb=[-1], input_ub=[1])
    np.random.seed(0)
    X, Y = MG.hjb_dataset(plant, 20, int(2 // plant.Ts))
    ls = [10.0 ** i for i in range(-2, 1)]
    with_bars("f14_hjb_cv", X, Y, 1, lambda: [R.KernelWrapper([l]) for l in ls], 500, dict(ls_grid=np.array(ls).reshape(-1, 1)))


if __name__ == "__main__":
    for a in sys.argv[1:] or ["duffing", "hjb"]:
        {"duffing": duffing, "hjb": hjb}[a]()
