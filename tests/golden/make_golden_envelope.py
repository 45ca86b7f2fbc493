#!/usr/bin/env python3
"""Per-unit parity bars for the ill-conditioned fixtures, produced by RUNNING THE REFERENCE in the build container.

For an ill-conditioned regularised system no two backward-stable solvers return the same solution, and LAPACK's gelsd
(what scipy.linalg.lstsq calls, regressors.py:155,165) is one more of them: measured here, gelsy, Cholesky and a
truncated symmetric eigen-solve agree with EACH OTHER far better than any of them agrees with gelsd.  A GPU solver cannot
be closer to the reference than LAPACK's other drivers are, so the bar of a unit is built from two reference-side numbers:

  spread_u    how far the reference's own score moves when its inputs are perturbed by one part in 1e15
              (its reproducibility), and
  roworder_u  how far it moves when the SAME reference code sees the SAME samples in another row order (the fit is a sum
              over samples: mathematically the same fit, numerically another summation order -- the GPU's Gram
              contraction necessarily sums in its own order), and
  envelope_u  how far the score moves when the SAME reference code calls another LAPACK driver for the two solves
              (scipy.linalg.lstsq(..., lapack_driver='gelsy'), a Cholesky solve, a symmetric eigen-solve with gelsd's
              eps * sigma_max cut-off): the largest of the three deviations from the gelsd score.

The reference's source is not modified: scipy.linalg.lstsq is wrapped for the duration of a run (the same mechanism as
the rank-recording spy of make_golden_configs.py).

    python tests/golden/make_golden_envelope.py cloth     -> f7b_cloth_cv_envelope.npz   (405 units x 4 extra sweeps)
    python tests/golden/make_golden_envelope.py duffing   -> f12b_duffing_envelope.npz   (60 fits x 3 drivers)
    python tests/golden/make_golden_envelope.py hjb       -> f8b_hjb_envelope.npz        (config 2: operators)
    python tests/golden/make_golden_envelope.py gain      -> f6b_cloth_gain_envelope.npz (shipped LQR gain, seed 0)
"""
import os
import random
import sys
import time

import numpy as np
import scipy.linalg

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden_configs as MG  # noqa: E402  (imports the reference)
import make_golden_duffing as MD  # noqa: E402

R, DS = MG.R, MG.DS
EPS = np.finfo(np.float64).eps


class LstsqSwap:
    """scipy.linalg.lstsq replaced by another LAPACK route for the duration of the block (a, b as the reference passes them)."""

    def __init__(self, mode):
        self.mode = mode
        self._orig = scipy.linalg.lstsq

    def __enter__(self):
        orig, mode = self._orig, self.mode

        def alt(a, b, *args, **kw):
            if mode == "gelsy":
                return orig(a, b, lapack_driver="gelsy")
            if mode == "chol":
                try:
                    return (scipy.linalg.cho_solve(scipy.linalg.cho_factor(a), b), None, a.shape[0], None)
                except np.linalg.LinAlgError:  # not positive definite to working precision: LU
                    return (scipy.linalg.solve(a, b), None, a.shape[0], None)
            if mode == "eigh":
                w, V = np.linalg.eigh((a + a.T) / 2)
                keep = w > EPS * w.max()
                return ((V[:, keep] / w[keep]) @ (V[:, keep].T @ b), None, int(keep.sum()), None)
            raise ValueError(mode)
        scipy.linalg.lstsq = alt
        return self

    def __exit__(self, *exc):
        scipy.linalg.lstsq = self._orig


def cloth():
    from sklearn.model_selection import GridSearchCV
    raw = MG.cloth_raw(range(10))
    X, Y = MG.data_matrices(raw)
    ls_grid = [(10.0 ** i, 10.0 ** j, 10.0 ** k) for i in range(3) for j in range(3) for k in range(3)]
    kernels = [R.ThreeDimensionalKernel(*c, 192) for c in ls_grid]
    gammas = np.power(10.0, np.arange(-7, -4))

    def sweep(Xs):
        clf = GridSearchCV(R.KoopmanNystromRegressor(6), {"kernel": kernels, "gamma": gammas, "m": [500]},
                           scoring="neg_root_mean_squared_error", n_jobs=1, refit=False)
        np.random.seed(42)
        clf.fit(Xs, Y)
        res = clf.cv_results_
        return np.stack([res[f"split{k}_test_score"] for k in range(5)], axis=1)

    g = np.load(f"{OUT}/f7_cloth_cv_full.npz")
    out = {}
    t0 = time.time()
    base = sweep(X)
    assert np.array_equal(base, g["split_scores"]), "the gelsd sweep is not the one stored in f7"
    print("cloth: gelsd sweep reproduced f7 bit for bit (%.0f s)" % (time.time() - t0), flush=True)
    prng = np.random.default_rng(11)
    out["scores_perturbed"] = sweep(X * (1 + 1e-15 * prng.standard_normal(X.shape)))
    print("cloth: perturbed sweep done (%.0f s)" % (time.time() - t0), flush=True)
    for mode in ("gelsy", "chol", "eigh"):
        with LstsqSwap(mode):
            out[f"scores_{mode}"] = sweep(X)
        print(f"cloth: {mode} sweep done (%.0f s)" % (time.time() - t0), flush=True)
    rel = lambda s: np.abs(s - base) / np.abs(base)
    out["spread"] = rel(out["scores_perturbed"])
    out["envelope"] = np.max(np.stack([rel(out[f"scores_{m}"]) for m in ("gelsy", "chol", "eigh")]), axis=0)
    for gv in gammas:
        sel = np.isclose(g["order_gamma"], gv, rtol=1e-6)
        print("gamma %.0e: spread max %.2e median %.2e | envelope max %.2e median %.2e" %
              (gv, out["spread"][sel].max(), np.median(out["spread"][sel]), out["envelope"][sel].max(),
               np.median(out["envelope"][sel])))
    np.savez_compressed(f"{OUT}/f7b_cloth_cv_envelope.npz", **out)


def duffing():
    g = np.load(f"{OUT}/f12_duffing_full.npz")
    X, Y, ms = g["X"], g["Y"], g["ms"]
    ref = g["ref_rmse"]
    dev = {}
    op_dev = {10: 0.0, 48: 0.0, 200: 0.0}  # operators of (seed 0, m): largest deviation of A, B, C over the drivers
    relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    t0 = time.time()
    for mode in ("gelsy", "chol", "eigh"):
        rm = np.zeros_like(ref)
        with LstsqSwap(mode):
            for si, seed in enumerate(g["seeds"]):
                for k, m in enumerate(ms):
                    reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=float(g["gamma"]), m=int(m))
                    reg.nystrom_centers_output = Y.T[:, g[f"idx_{seed}_{k}"]]
                    reg.fit(X, Y)
                    rm[si, k] = MD.validate_dyn_sys(reg, g[f"traj_{seed}"], g[f"ctrl_{seed}"])[0]
                    if seed == 0 and int(m) in op_dev:
                        op_dev[int(m)] = max(op_dev[int(m)], relf(reg.A, g[f"A_m{m}"]), relf(reg.B, g[f"B_m{m}"]),
                                             relf(reg.C, g[f"C_m{m}"]))
        dev[mode] = np.abs(rm - ref) / ref
        print(f"duffing: {mode} done (%.0f s), max deviation from gelsd %.2e" % (time.time() - t0, dev[mode].max()), flush=True)
    # the LQR chain of benchmark_lqr_classic.py:256-299 (m = 20) through the other drivers
    ds = MD.duffing_plant()
    x0, reference, steps = np.array([-0.5, 0.0]).reshape([-1, 1]), np.zeros((2, 1)), int(g["lqr_steps"])
    lqr_env = {int(seed): np.zeros(4) for seed in g["seeds"]}
    for mode in ("gelsy", "chol", "eigh"):
        with LstsqSwap(mode):
            for seed in g["seeds"]:
                reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=float(g["gamma"]), m=20)
                reg.nystrom_centers_output = Y.T[:, g[f"lqr_idx_{seed}"]]
                reg.fit(X, Y)
                K, _, us, st = MD.closed_loop(ds, reg, x0, reference, steps)
                d4 = np.array([max(relf(reg.A, g[f"lqr_A_{seed}"]), relf(reg.B, g[f"lqr_B_{seed}"]), relf(reg.C, g[f"lqr_C_{seed}"])),
                               relf(K, g[f"lqr_K_{seed}"]), relf(us, g[f"lqr_us_{seed}"]), relf(st, g[f"lqr_states_{seed}"])])
                lqr_env[int(seed)] = np.maximum(lqr_env[int(seed)], d4)
    print("duffing LQR envelope (operators, K, controls, states) by seed:", lqr_env, flush=True)
    # the reference itself (gelsd) on the same samples in another row order: sweep, operators of (seed 0; 10, 48, 200), LQR chain
    perm = np.random.default_rng(5).permutation(X.shape[0])
    Xr, Yr = np.ascontiguousarray(X[perm]), np.ascontiguousarray(Y[perm])
    rm_r = np.zeros_like(ref)
    op_r = {10: 0.0, 48: 0.0, 200: 0.0}
    for si, seed in enumerate(g["seeds"]):
        for k, m in enumerate(ms):
            reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=float(g["gamma"]), m=int(m))
            reg.nystrom_centers_output = Y.T[:, g[f"idx_{seed}_{k}"]]
            reg.fit(Xr, Yr)
            rm_r[si, k] = MD.validate_dyn_sys(reg, g[f"traj_{seed}"], g[f"ctrl_{seed}"])[0]
            if seed == 0 and int(m) in op_r:
                op_r[int(m)] = max(relf(reg.A, g[f"A_m{m}"]), relf(reg.B, g[f"B_m{m}"]), relf(reg.C, g[f"C_m{m}"]))
    roworder = np.abs(rm_r - ref) / ref
    lqr_row = {}
    for seed in g["seeds"]:
        reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=float(g["gamma"]), m=20)
        reg.nystrom_centers_output = Y.T[:, g[f"lqr_idx_{seed}"]]
        reg.fit(Xr, Yr)
        K, _, us, st = MD.closed_loop(ds, reg, x0, reference, steps)
        lqr_row[int(seed)] = np.array([max(relf(reg.A, g[f"lqr_A_{seed}"]), relf(reg.B, g[f"lqr_B_{seed}"]), relf(reg.C, g[f"lqr_C_{seed}"])),
                                       relf(K, g[f"lqr_K_{seed}"]), relf(us, g[f"lqr_us_{seed}"]), relf(st, g[f"lqr_states_{seed}"])])
    print("duffing: reference in another row order: rmse max rel %.2e; operators %s; LQR %s" % (roworder.max(), op_r, lqr_row), flush=True)
    env = np.max(np.stack(list(dev.values())), axis=0)
    print("duffing operator envelope (seed 0):", op_dev)
    np.savez_compressed(f"{OUT}/f12b_duffing_envelope.npz", envelope=env, op_envelope_m=np.array(sorted(op_dev)),
                        op_envelope=np.array([op_dev[k] for k in sorted(op_dev)]), roworder=roworder,
                        op_roworder=np.array([op_r[k] for k in sorted(op_r)]),
                        **{f"lqr_roworder_{k}": v for k, v in lqr_row.items()},
                        **{f"lqr_envelope_{k}": v for k, v in lqr_env.items()}, **{f"dev_{k}": v for k, v in dev.items()})
    print("duffing envelope by m (max over seeds):", dict(zip(ms.tolist(), np.round(env.max(axis=0), 6).tolist())))


def hjb():
    """Config 2 (f8: HJB N = 1e4, m = 200, Matern-5/2): how far the reference's operators move with another LAPACK driver for
    the two solves and with the samples in another row order (the 1e-15 input perturbation is already in f8)."""
    g = np.load(f"{OUT}/f8_hjb_config2.npz")
    X, Y, idx = g["X"], g["Y"], g["idx"]
    ls, gamma, m = float(g["ls"]), float(g["gamma"]), int(g["m"])
    relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))

    def fit(Xs, Ys):
        reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([ls]), gamma=gamma, m=m)
        reg.nystrom_centers_output = Y.T[:, idx]
        reg.fit(Xs, Ys)
        return np.array([relf(reg.A, g["A"]), relf(reg.B, g["B"]), relf(reg.C, g["C"]), relf(reg.weights, g["W"])])

    env = np.zeros(4)
    for mode in ("gelsy", "chol", "eigh"):
        with LstsqSwap(mode):
            dev = fit(X, Y)
        print("hjb:", mode, "moves (A, B, C, W) by", dev, flush=True)
        env = np.maximum(env, dev)
    perm = np.random.default_rng(5).permutation(X.shape[0])
    row = fit(np.ascontiguousarray(X[perm]), np.ascontiguousarray(Y[perm]))
    print("hjb: another row order moves (A, B, C, W) by", row, "; 1e-15 input perturbation (f8):", float(g["op_sensitivity"]))
    np.savez_compressed(f"{OUT}/f8b_hjb_envelope.npz", op_envelope=env, op_roworder=row)


def gain():
    """The authors' shipped K_lqr_seed_0.csv (f6): how well the reference code run HERE reproduces it (gelsd, the reference as
    it is), and how far the gain moves when its two solves use another LAPACK driver -- the bar of the build's gain.
    benchmark_lqr_cloth.py:218-263: seed 0, m = 100, kernel l = (10, 10, 10), gamma = 1e-7, c = 0.005, Q = c C'C, R = I; the
    DARE through scipy (control.dlqr is not installed: make_golden_configs.py)."""
    g = np.load(f"{OUT}/f6_cloth_known_gain.npz")
    tr, u = g["trajs"], g["inputs"]
    X = np.ascontiguousarray(np.hstack([np.vstack((tr[i][:, :-1], u[i][:, :-1])) for i in range(30)]).T)
    Y = np.ascontiguousarray(np.hstack([tr[i][:, 1:] for i in range(30)]).T)
    relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))

    def K_of(Xs, Ys):
        np.random.seed(0)
        reg = R.KoopmanNystromRegressor(6, kernel=R.ThreeDimensionalKernel(10, 10, 10, 192), gamma=1e-7, m=100)
        reg.fit(Xs, Ys)
        Q = 0.005 * reg.C.T @ reg.C
        Q = (Q + Q.T) / 2
        P = scipy.linalg.solve_discrete_are(reg.A, reg.B, Q, np.eye(6))
        K = np.linalg.solve(reg.B.T @ P @ reg.B + np.eye(6), reg.B.T @ P @ reg.A)
        return K[[0, 3, 1, 4, 2, 5], :], reg
    K0, reg0 = K_of(X, Y)
    out = dict(K_reference_here=K0, reproduces_shipped=relf(K0, g["K_lqr_seed_0"]))
    print("gain: the reference run here against the shipped CSV:", out["reproduces_shipped"], flush=True)
    env = 0.0
    for mode in ("gelsy", "chol", "eigh"):
        with LstsqSwap(mode):
            Km, regm = K_of(X, Y)
        dev = relf(Km, K0)
        print(f"gain: {mode} moves K by {dev:.3e} (A by {relf(regm.A, reg0.A):.3e})", flush=True)
        out[f"K_{mode}"] = dev
        env = max(env, dev)
    rng = np.random.default_rng(3)
    Kp, _ = K_of(X * (1 + 1e-15 * rng.standard_normal(X.shape)), Y * (1 + 1e-15 * rng.standard_normal(Y.shape)))
    out["K_spread"] = relf(Kp, K0)
    out["K_envelope"] = env
    print("gain: 1e-15 input perturbation moves K by", out["K_spread"], "; envelope", env)
    np.savez_compressed(f"{OUT}/f6b_cloth_gain_envelope.npz", **out)


if __name__ == "__main__":
    for a in sys.argv[1:] or ["duffing", "cloth", "hjb", "gain"]:
        {"cloth": cloth, "duffing": duffing, "hjb": hjb, "gain": gain}[a]()
