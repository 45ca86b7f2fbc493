"""GPU parity tests at the sizes BASELINE.json's configs name:

  config 3  the real cloth hyper-parameter sweep (n = 1010, m = 500, 27 kernels x 3 gammas x 5 folds = 405 fits) against
            per-fold scores that scikit-learn's GridSearchCV produced driving the REFERENCE estimator (f7);
  config 2  HJB regenerated at N = 1e4, m = 200, Matern-5/2: Nystrom operators, forecasts and plant-in-the-loop LQR
            controls against the reference, and Nystrom-vs-exact-kernel forecasts as benchmark_lqr_hjb.py:313,378 (f8);
  C5        n = 1e6, m = 8000, d = 1024 in fp64 on one GPU through size-independent properties computed on the device,
            plus reference parity on a scaled twin n = 2e4, m = 1024, d = 1024 with a 20-step forecast (f11).
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

from conftest import relf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nk():
    import nys_koop_lqr_amd as nk
    nk.get_context()
    return nk


@pytest.fixture(scope="module")
def O():
    from oracle import nk_oracle
    return nk_oracle


# ---------------------------------------------------------------------------------------------------------------
# config 3: the cloth CV sweep at its real shape
# ---------------------------------------------------------------------------------------------------------------
def _cloth_cv_inputs(golden):
    g = golden("f7_cloth_cv_full.npz")
    t = golden("cloth_trajs_all.npz")
    states = t["states_e10"] / 1e10
    X = np.hstack([np.vstack((states[i][:, :-1], t["inputs"][i][:, :-1])) for i in range(10)]).T
    Y = np.hstack([states[i][:, 1:] for i in range(10)]).T
    return g, np.ascontiguousarray(X), np.ascontiguousarray(Y)


def test_cloth_cv_grid_full_shape_vs_reference_gridsearch(nk, golden):
    """benchmark_lqr_cloth.py:39-66,157-159 with the landmark draws replayed from the seed in GridSearchCV's order.

    What can agree, by candidate class (measured with LAPACK alone, tools/gelsd_truncation_study.py): at gamma = 1e-7 the
    regularised systems have sigma_min / sigma_max <= 3e-16; the reference's gelsd rank-truncates 8 of the 27 kernels
    there (up to 360 of 506 singular values, all within a factor 2 of eps * sigma_max, i.e. chosen by rounding noise) and
    LAPACK's own SVD with the same cut-off, gelsy and Cholesky all sit 1e-3..4e-3 from gelsd's score (and 1e-4 from each
    other).  At gamma = 1e-6 the spread is 1e-4, at gamma = 1e-5 it is 1e-5.  The bars below are those spreads with a
    margin; the ranking of the candidates and the selected hyper-parameters must be the reference's."""
    from nys_koop_lqr_amd import harness
    g, X, Y = _cloth_cv_inputs(golden)
    assert X.shape == (1010, 198)
    cands = []
    for c in range(len(g["order_gamma"])):
        ls = g["ls_grid"][int(g["order_kernel"][c])]
        cands.append(dict(kernel=nk.ThreeDimensionalKernel(*ls, 192), gamma=float(g["order_gamma"][c]), m=int(g["m"])))
    np.random.seed(int(g["seed"]))
    t0 = time.perf_counter()
    res = harness.grid_search_cv(X, Y, 6, cands, n_splits=5, batch=32, batch_groups=2)  # lock-step batched (nk_cv_grid)
    dt = time.perf_counter() - t0
    # the batched sweep computes the very bits of the one-unit-at-a-time path: replay the first three candidates unbatched
    np.random.seed(int(g["seed"]))
    ref3 = harness.grid_search_cv(X, Y, 6, cands[:3], n_splits=5)
    assert np.array_equal(ref3["split_scores"], res["split_scores"][:3])
    sc, ref = res["split_scores"], g["split_scores"]
    assert sc.shape == ref.shape == (81, 5) and np.all(np.isfinite(sc))
    rel = np.abs(sc - ref) / np.abs(ref)
    gam = g["order_gamma"]
    truncated = (g["lstsq_rank"] < g["lstsq_size"]).any(axis=(1, 2))
    # Unit by unit (405 bars instead of three): two reference-side numbers from tests/golden/make_golden_envelope.py --
    # `spread`, how far the reference's own score moves under a 1e-15 perturbation of its inputs, and `envelope`, how far
    # it moves when the SAME reference code calls another LAPACK driver (gelsy, Cholesky, eigen-solve with gelsd's
    # cut-off) for its two solves.  bar_u = max(10 spread_u, 3 envelope_u, 1e-7), the multipliers of the Duffing and HJB
    # configs (the build also forms the systems differently: summation order, polar iteration for the square root); the
    # floor (a tenth of the north-star tolerance) carries the well-conditioned units, which the reference reproduces to
    # 1e-10 and this build to 1e-8.  Measured: worst unit at 0.72 of its bar (3.8e-7 against an envelope of 1.7e-7).
    e = golden("f7b_cloth_cv_envelope.npz")
    bar = np.maximum(np.maximum(10.0 * e["spread"], 3.0 * e["envelope"]), 1e-7)
    ratio = rel / bar
    worst = np.unravel_index(np.argmax(ratio), ratio.shape)
    report = {}
    for gv in (1e-7, 1e-6, 1e-5):
        sel = np.isclose(gam, gv, rtol=1e-6)
        report[gv] = dict(max_err=float(rel[sel].max()), median_err=float(np.median(rel[sel])),
                          max_err_over_bar=float(ratio[sel].max()), envelope_max=float(e["envelope"][sel].max()),
                          spread_max=float(e["spread"][sel].max()))
    print(f"\n[cloth CV 405 units] {dt:.2f} s = {405 / dt:.0f} units/s; by gamma: {report}; reference truncated "
          f"{int(truncated.sum())} candidates; worst unit (candidate {worst[0]}, fold {worst[1]}): err {rel[worst]:.2e}, "
          f"spread {e['spread'][worst]:.2e}, envelope {e['envelope'][worst]:.2e}")
    assert ratio.max() <= 1.0, (worst, float(rel[worst]), float(bar[worst]))
    # the selection is the reference's
    assert res["best_index"] == int(np.argmax(g["mean_test_score"]))
    mean_rel = np.abs(res["mean_test_score"] - g["mean_test_score"]) / np.abs(g["mean_test_score"])
    assert mean_rel.max() < 1e-2
    top_ref = set(np.argsort(-g["mean_test_score"])[:5].tolist())
    top_got = set(np.argsort(-res["mean_test_score"])[:5].tolist())
    assert len(top_ref & top_got) >= 4


# ---------------------------------------------------------------------------------------------------------------
# configs 1 and 2: the hyper-parameter searches (learn_hyperparams of benchmark_lqr_classic.py:44-64, benchmark_lqr_hjb.py:47-70)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,units", [("f13_duffing_cv.npz", 80), ("f14_hjb_cv.npz", 240)])
def test_duffing_and_hjb_hyperparameter_searches_vs_reference_gridsearch(nk, golden, name, units):
    """The (kernel, gamma) x 5-fold searches of configs 1 and 2 at their real shape (n = 3980 validation rows, m = 500, 16 gammas;
    1 / 3 Matern kernels) against scikit-learn's GridSearchCV driving the reference estimator (f13 / f14,
    tests/golden/make_golden_cv.py): every (candidate, fold) score within max(10 x the reference's own movement under a 1e-15
    input perturbation, 3 x its movement with another LAPACK driver, 1e-7), the same best candidate; lock-step batched."""
    from nys_koop_lqr_amd import harness
    g = golden(name)
    X, Y = g["X"], g["Y"]
    assert X.shape[0] == 3980 and g["split_scores"].size == units  # 20 x int(2 // 0.01) = 20 x 199 rows, as the reference draws them
    cands = [dict(kernel=nk.KernelWrapper(g["ls_grid"][int(k)]), gamma=float(gm), m=int(g["m"]))
             for k, gm in zip(g["order_kernel"], g["order_gamma"])]
    np.random.seed(int(g["seed"]))
    t0 = time.perf_counter()
    res = harness.grid_search_cv(X, Y, 1, cands, n_splits=5, batch=16, batch_groups=2)
    dt = time.perf_counter() - t0
    sc, ref = res["split_scores"], g["split_scores"]
    assert sc.shape == ref.shape and np.all(np.isfinite(sc))
    rel = np.abs(sc - ref) / np.abs(ref)
    bar = np.maximum(np.maximum(10.0 * g["spread"], 3.0 * g["envelope"]), 1e-7)
    worst = np.unravel_index(np.argmax(rel / bar), rel.shape)
    print(f"\n[{name}] {units} units in {dt:.2f} s = {units / dt:.0f} units/s; score error max {rel.max():.2e} median "
          f"{np.median(rel):.2e}; worst unit {worst}: err {rel[worst]:.2e} = {float((rel / bar)[worst]):.2f} of its bar (spread "
          f"{g['spread'][worst]:.2e}, envelope {g['envelope'][worst]:.2e})")
    assert (rel / bar).max() <= 1.0, (worst, float(rel[worst]), float(bar[worst]))
    assert res["best_index"] == int(np.argmax(g["mean_test_score"]))


# ---------------------------------------------------------------------------------------------------------------
# config 2: HJB, N = 1e4, m = 200, Nystrom vs exact kernel
# ---------------------------------------------------------------------------------------------------------------
def test_hjb_config2_nystrom_vs_exact_kernel(nk, O, golden):
    from nys_koop_lqr_amd import harness
    g = golden("f8_hjb_config2.npz")
    X, Y, idx = g["X"], g["Y"], g["idx"]
    assert X.shape == (10000, 2) and int(g["m"]) == 200
    ls, gamma = float(g["ls"]), float(g["gamma"])
    reg = nk.KoopmanNystromRegressor(1, kernel=nk.KernelWrapper([ls]), gamma=gamma, m=200)
    reg.nystrom_centers_output = Y.T[:, idx]
    reg.fit(X, Y)
    errs = dict(A=relf(reg.A, g["A"]), B=relf(reg.B, g["B"]), C=relf(reg.C, g["C"]), W=relf(reg.weights, g["W"]),
                predict=relf(reg.predict(g["Xq"]), g["nys_predict"]))
    print("\n[HJB N=1e4 m=200] operator errors vs reference:", errs)
    assert errs["predict"] < 1e-6
    # the operators of this fit are ill-determined (cond(inner) = 1.3e13).  The bar per operator is what the REFERENCE itself
    # reproduces (f8b, tests/golden/make_golden_envelope.py hjb): the larger of (a) its move when the training rows are
    # given in another order -- another summation order in the Gram products, which is what a GPU has -- and (b) its move
    # when scipy's lstsq driver (gelsd) is swapped for gelsy / Cholesky / eigh; times 3.  (Round 2 needed 200 x the 1e-15
    # input sensitivity here: the blocked Cholesky solve multiplied by explicitly inverted diagonal blocks, a backward error
    # of 6 eps instead of LAPACK's 0.3 eps, and times cond that was the whole gap; every such product now takes a
    # correction step from the data -- DESIGN.md section 3.)
    env = golden("f8b_hjb_envelope.npz")
    bars = 3.0 * np.maximum(env["op_roworder"], env["op_envelope"])
    print("[HJB N=1e4 m=200] bars (A, B, C, W):", bars)
    for nm, bar in zip("ABCW", bars):
        assert errs[nm] < bar, (nm, errs[nm], bar)
    # open-loop forecasts of benchmark_lqr_hjb.py:23-44 on the seeded test trajectories (relative-% RMSE, :42)
    trajs, ctrls = g["test_trajs"], g["test_controls"]
    nys_rmse = []
    for k in range(trajs.shape[0]):
        sim = harness.open_loop_forecast(reg, trajs[k], ctrls[k])
        assert relf(sim, g["nys_forecasts"][k]) < 1e-5
        nys_rmse.append(harness.validate_dyn_sys(reg, trajs[k], ctrls[k], relative=True))
    assert np.allclose(nys_rmse, g["nys_rmse"], rtol=1e-4)
    # plant-in-the-loop LQR against the analytic optimum (:73-97, :296-313)
    plant = lambda x, u: O.hjb_step(x, u, 0.01)
    K = reg.solve_lqr(Q=reg.C.T @ reg.C, R=np.eye(1))
    assert relf(K, g["K"]) < 1e-4
    steps = int(g["cl_steps"])
    xs, us = harness.lqr_control_plant(steps, np.array([[0.0]]), np.array([[0.9]]), reg, K, plant)
    assert relf(us.squeeze(), g["cl_u"]) < 1e-5 and relf(xs, g["cl_x"]) < 1e-6
    u_opt = O.hjb_optimal_controls(plant, 0.9, steps)
    assert relf(u_opt, g["u_opt"]) < 1e-12
    rc = harness.control_rmse_percent(us, u_opt)
    assert abs(rc - float(g["rmse_control"])) < 1e-3 * float(g["rmse_control"])
    # exact-kernel comparator (regressors.py:58-111) at the reference's own N = 4000
    Ne = int(g["exact_N"])
    kreg = nk.KoopmanKernelRegressor(1, kernel=nk.KernelWrapper([ls]), gamma=gamma)
    kreg.fit(X[:Ne], Y[:Ne])
    assert relf(kreg.predict(g["Xq"]), g["exact_predict"]) < 1e-6
    ex_rmse = []
    for k in range(trajs.shape[0]):
        sim = harness.open_loop_forecast(kreg, trajs[k], ctrls[k])
        assert relf(sim, g["exact_forecasts"][k]) < 1e-4
        ex_rmse.append(np.sqrt(np.sum(np.square(trajs[k] - sim))) / np.sqrt(np.sum(np.square(sim))) * 100)
    assert np.allclose(ex_rmse, g["exact_rmse"], rtol=1e-3)
    # Nystrom vs exact on forecasts: the gap the reference reports is reproduced
    gap_ref = g["nys_rmse"] - g["exact_rmse"]
    gap = np.array(nys_rmse) - np.array(ex_rmse)
    print("[HJB] forecast %-RMSE Nystrom", np.round(nys_rmse, 4), "exact", np.round(ex_rmse, 4))
    assert np.allclose(gap, gap_ref, atol=1e-3 * np.max(np.abs(g["nys_rmse"])))


# ---------------------------------------------------------------------------------------------------------------
# C5: scaled twin against the reference, full size through properties
# ---------------------------------------------------------------------------------------------------------------
def _c5_like(n, d, p, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((n, d)).astype(np.float32)
    U = rng.standard_normal((n, p)).astype(np.float32)
    Wt = (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d)).astype(np.float32)
    Bt = (rng.standard_normal((p, d)) * 0.1).astype(np.float32)
    Y = (np.tanh(S.astype(np.float64) @ Wt) + U.astype(np.float64) @ Bt).astype(np.float32).astype(np.float64)
    X = np.hstack([S, U]).astype(np.float64)
    return X, Y, rng


def test_c5_scaled_twin_vs_reference(nk, golden):
    g = golden("f11_c5_twin.npz")
    n, d, p, m = int(g["n"]), int(g["d"]), int(g["p"]), int(g["m"])
    X, Y, rng = _c5_like(n, d, p, int(g["seed"]))
    assert np.array_equal(X[7], g["x_check"]) and np.array_equal(Y[7], g["y_check"])  # the recipe regenerates the inputs
    ls, gamma = float(g["ls"]), float(g["gamma"])
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=gamma, m=m)
    reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, g["idx"]])
    reg.fit(X, Y)
    prng = np.random.default_rng(int(g["probe_seed"]))
    PA = prng.standard_normal((m, 16))
    PC = prng.standard_normal((m, 16))
    errs = dict(A=relf(reg.A @ PA, g["A_probe"]), At=relf(reg.A.T @ PA, g["At_probe"]), C=relf(reg.C @ PC, g["C_probe"]),
                B=relf(reg.B, g["B"]), predict=relf(reg.predict(X[g["q"]]), g["predict"]))
    Useq = g["Useq"]
    sim = reg.rollout(X[int(g["x0_row"]), :d], Useq)
    errs["forecast20"] = relf(sim, g["forecast"])
    print("\n[C5 twin n=2e4 m=1024 d=1024] errors vs reference:", errs, "sqrt iters", reg.fit_stats_["sqrt_iters"])
    assert abs(np.linalg.norm(reg.A) - float(g["A_fro"])) < 1e-6 * float(g["A_fro"])
    assert max(errs["A"], errs["At"], errs["B"], errs["C"], errs["predict"]) < 1e-6
    assert errs["forecast20"] < 1e-5


def test_c5_scaled_twin_fp32_engine(nk, golden):
    """BASELINE.json configs[4] names fp32 for the stress configuration.  The fp32 engine (kernel blocks and Gram
    contractions on the fp32 matrix pipe; Gram accumulators and everything m x m in fp64) on the same twin, against the
    same reference outputs: fp32 cannot reach the 1e-6 operator bar of the fp64 path -- what it does
    reach on operators is printed, and the quantities the stress configuration is graded on (predictions, the 20-step
    forecast) are asserted."""
    g = golden("f11_c5_twin.npz")
    n, d, p, m = int(g["n"]), int(g["d"]), int(g["p"]), int(g["m"])
    X, Y, rng = _c5_like(n, d, p, int(g["seed"]))
    ls, gamma = float(g["ls"]), float(g["gamma"])
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=gamma, m=m)
    reg.compute_dtype = "f32"
    reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, g["idx"]])
    reg.fit(X, Y)
    st = reg.fit_stats_
    prng = np.random.default_rng(int(g["probe_seed"]))
    PA = prng.standard_normal((m, 16))
    PC = prng.standard_normal((m, 16))
    errs = dict(A=relf(reg.A @ PA, g["A_probe"]), At=relf(reg.A.T @ PA, g["At_probe"]), C=relf(reg.C @ PC, g["C_probe"]),
                B=relf(reg.B, g["B"]), predict=relf(reg.predict(X[g["q"]]), g["predict"]))
    errs["forecast20"] = relf(reg.rollout(X[int(g["x0_row"]), :d], g["Useq"]), g["forecast"])
    print("\n[C5 twin, fp32 engine] errors vs reference:", errs, "| kmat %.2f ms, gram %.2f ms" % (st["ms_kmat"], st["ms_gram"]))
    # measured: operators 4-6e-2, predictions 1.0e-2, 20-step forecast 7e-2.  This twin is hard on fp32 by construction: in
    # d = 1024 all points are nearly equidistant, every kernel value is 0.37 +- 0.015, and the fit lives on the +- 0.015 --
    # two digits of every fp32 operand carry no information (the fp64 engine's own errors on it are 1e-8)
    assert errs["predict"] < 3e-2 and errs["forecast20"] < 2e-1 and errs["B"] < 2e-3


@pytest.mark.timeout(900)
def test_c5_full_size_properties(nk):
    """n = 1e6, m = 8000, d = 1024, p = 6, fp64 end to end, inputs generated on the device (torch is plumbing: device
    memory and a random generator).  No CPU reference exists at this size (the reference would need ~1 day), so the fit
    is pinned through identities the reference's algebra implies, evaluated on the device through the library's own
    GEMM entry point:
        S S = K_mm + jitter I,   S^-1 S = I                       (regressors.py:139-140)
        W = C [A B]                                               (:167)
        additivity: the Gram accumulators of [0, n/2) + [n/2, n) equal those of all rows   (:151,153,162,164)
        normal equations on a row sample: with sol = inner^-1 right, [A B] = S^-1 cross sol -- checked in the form
        [A B] blkdiag(K S^-1, I)^-1 inner = S^-1 cross on random probe vectors (:151-156).
    """
    import torch
    from nys_koop_lqr_amd import _lib
    n, m, d, p = 1_000_000, 8000, 1024, 6
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    Xd = torch.empty((n, d + p), dtype=torch.float64, device=dev)
    Yd = torch.empty((n, d), dtype=torch.float64, device=dev)
    Wt = torch.randn((d, d), generator=gen, dtype=torch.float64, device=dev) * (0.9 / np.sqrt(d))
    Bt = torch.randn((p, d), generator=gen, dtype=torch.float64, device=dev) * 0.1
    step = 100_000
    for r in range(0, n, step):
        blk = torch.randn((step, d + p), generator=gen, dtype=torch.float64, device=dev)
        Xd[r:r + step] = blk
        Yd[r:r + step] = torch.tanh(blk[:, :d] @ Wt) + blk[:, d:] @ Bt
    del blk
    torch.cuda.synchronize()
    idx = np.random.RandomState(0).choice(n, m, replace=False)
    Z = Yd[torch.from_numpy(idx).to(dev)].cpu().numpy()
    ls, gamma, jitter = 32.0, 1e-6, 1e-6
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=gamma, m=m)
    reg.nystrom_centers_output = np.ascontiguousarray(Z.T)
    t0 = time.perf_counter()
    reg.fit(Xd, Yd)
    A, B, Cm, W = reg.A, reg.B, reg.C, reg.weights
    dt = time.perf_counter() - t0
    st = reg.fit_stats_
    print(f"\n[C5 full size] fit {dt:.2f} s wall, {st['ms_total']:.0f} ms device; Gram launches {st['gram_kernel_launches']}, "
          f"sqrt iters {st['sqrt_iters']}, ranks {st['rank_inner']}/{st['rank_inner_rec']}")
    assert np.all(np.isfinite(A)) and np.all(np.isfinite(Cm))
    ctx = nk.get_context()

    def dgemm(Am, Bm, ta=False, tb=False):
        Am, Bm = np.ascontiguousarray(Am), np.ascontiguousarray(Bm)
        M = Am.shape[1] if ta else Am.shape[0]
        K = Am.shape[0] if ta else Am.shape[1]
        N = Bm.shape[0] if tb else Bm.shape[1]
        out = np.empty((M, N))
        _lib.check(ctx.lib.nk_gemm(ctx.handle, int(ta), int(tb), M, N, K, 1.0, Am.ctypes.data, Am.shape[1], Bm.ctypes.data,
                                   Bm.shape[1], 0.0, out.ctypes.data, N))
        return out

    def get(which, shape):
        out = np.empty(shape)
        _lib.check(ctx.lib.nk_model_get(ctx.handle, reg._model, which.encode(), out.ctypes.data, shape[1]))
        return out

    S, Si = get("S", (m, m)), get("I", (m, m))
    Kmm = reg.kernel.kernel(Z, Z)
    Kj = Kmm + jitter * np.eye(m)
    assert relf(dgemm(S, S), Kj) < 1e-9
    assert relf(dgemm(Si, S), np.eye(m)) < 1e-8
    assert relf(W, dgemm(Cm, np.hstack((A, B)))) < 1e-10
    # additivity of the Gram accumulators over row ranges (packed block, on the device)
    cnt = reg.gram_size(d)
    g_all = torch.empty(cnt, dtype=torch.float64, device=dev)
    g_a = torch.empty(cnt, dtype=torch.float64, device=dev)
    g_b = torch.empty(cnt, dtype=torch.float64, device=dev)
    reg.gram_partial(Xd, Yd, out=g_all)
    reg.gram_partial(Xd, Yd, row_ranges=[(0, n // 2)], out=g_a)
    reg.gram_partial(Xd, Yd, row_ranges=[(n // 2, n)], out=g_b)
    add_err = float(torch.linalg.vector_norm(g_a + g_b - g_all) / torch.linalg.vector_norm(g_all))
    assert add_err < 1e-12, add_err
    # normal equations: inner = G1 + gamma n blkdiag(Kj, I), cross = G2; [A B] R^-1 inner = S^-1 cross with
    # R = blkdiag(K S^-1, I) and K S^-1 = S - jitter S^-1   =>   tested on probes v:  [A B] (R^-1 (inner v)) = S^-1 (cross v)
    mp = m + p
    G = g_all.cpu().numpy()
    G1 = G[:mp * mp].reshape(mp, mp)
    G2 = G[mp * mp:mp * mp + m * mp].reshape(m, mp)
    inner = G1.copy()
    inner[:m, :m] += gamma * n * Kj
    inner[m:, m:] += gamma * n * np.eye(p)
    V = np.random.default_rng(5).standard_normal((mp, 8))
    iv = dgemm(inner, V)
    KSi = S - jitter * Si
    rinv_top = np.linalg.solve(KSi, iv[:m])  # host solve of an m x m system with 8 right-hand sides: part of the checker
    lhs = dgemm(np.hstack((A, B)), np.vstack((rinv_top, iv[m:])))
    rhs = dgemm(Si, dgemm(G2, V))
    ne = relf(lhs, rhs)
    print(f"[C5 full size] additivity {add_err:.1e}, normal-equation residual {ne:.1e}")
    assert ne < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# config 1 at the reference driver's real shape: benchmark_lqr_classic.py:174-179,211-299 on duffing/*.csv (n = 69 900)
# ---------------------------------------------------------------------------------------------------------------
def _duffing_fit(nk, g, idx, m):
    reg = nk.KoopmanNystromRegressor(1, kernel=nk.KernelWrapper([1, 1]), gamma=float(g["gamma"]), m=int(m))
    reg.nystrom_centers_output = np.ascontiguousarray(g["Y"].T[:, idx])
    reg.fit(g["X"], g["Y"])
    return reg


def test_duffing_full_shape_open_loop_sweep_vs_reference(nk, golden):
    """The open-loop validation sweep of benchmark_lqr_classic.py:211-255 at its real shape -- n = 69 900 samples, Matern-5/2,
    m = around(logspace(1, 2.3, 20)), seeds 0..2: 60 fits on the HIP path with the landmarks the reference drew, each
    followed by the 100-step forecast and the relative-% RMSE of :39.

    Bar, fit by fit, from two reference-side numbers (tests/golden/make_golden_envelope.py): `spread` -- how far the
    reference's own RMSE moves when its inputs are perturbed by one part in 1e15 -- and `envelope` -- how far it moves when
    the SAME reference code calls another LAPACK driver for its two solves (gelsy, Cholesky, truncated symmetric
    eigen-solve).  For m >= 57 (cond(inner) ~ 1e13..1e14) gelsd is the odd one out: the three other drivers agree with
    each other to 1e-4 and sit 5e-5..2e-2 from gelsd, which is itself reproducible to 1e-6..3e-4 -- a systematic error of
    its divide-and-conquer SVD (absolute, not relative, accuracy of sigma_min ~ 1e-14 sigma_max), not noise.  No solver can
    be closer to the reference than LAPACK's own drivers are.  Third number, `roworder`: the reference on the same samples
    in another row order (another summation order of the same sum, which is what a GPU contraction is).
    bar = max(10 spread, 10 roworder, 3 envelope, 1e-8) -- 3 rather than 1 on the envelope because each LAPACK driver differs
    from the reference in the solver alone, this build in the solver AND in how the systems are formed (summation order
    of the Gram contraction, square root by a polar iteration instead of a Schur decomposition); observed worst: 1.8
    envelopes at (seed 0, m = 146), where cond(inner) ~ 1e14.  Below m = 57 the
    envelope is within the spread and the bar is the reference's reproducibility alone.  The first column (m = 10) is
    also held against the file the AUTHORS shipped (duffing/all_rmses_nystrom_double_dataset.csv, seeds 0..7)."""
    from nys_koop_lqr_amd import harness
    g = golden("f12_duffing_full.npz")
    ms = g["ms"]
    K_BAR, K_ENV, FLOOR = 10.0, 3.0, 1e-8
    ref, refp = g["ref_rmse"], g["ref_rmse_perturbed"]
    spread = np.abs(refp - ref) / ref
    e12 = golden("f12b_duffing_envelope.npz")
    envelope, roworder = e12["envelope"], e12["roworder"]
    rows = []
    worst = 0.0
    t0 = time.time()
    for si, seed in enumerate(g["seeds"]):
        traj, ctrl = g[f"traj_{seed}"], g[f"ctrl_{seed}"]
        for k, m in enumerate(ms):
            reg = _duffing_fit(nk, g, g[f"idx_{seed}_{k}"], m)
            rmse = harness.validate_dyn_sys(reg, traj, ctrl, relative=True)
            err = abs(rmse - ref[si, k]) / ref[si, k]
            bar = max(K_BAR * spread[si, k], K_BAR * roworder[si, k], K_ENV * envelope[si, k], FLOOR)
            rows.append((int(seed), int(m), ref[si, k], rmse, err, spread[si, k], envelope[si, k], err / bar))
            worst = max(worst, err / bar)
    dt = time.time() - t0
    print(f"\nduffing full shape: 60 fits + forecasts at n = 69900 in {dt:.2f} s; worst err / bar = {worst:.3f}")
    print("seed    m   ref_rmse%      gpu_rmse%     rel.err   ref.spread   envelope  err/bar")
    for r in rows:
        print("%4d %4d %12.6f %12.6f %10.2e %10.2e %10.2e %8.3f" % r)
    assert worst <= 1.0, [r for r in rows if r[-1] > 1.0]
    # the authors' own numbers: first column of the shipped file, seeds 0..7
    first = []
    for seed in range(8):
        reg = _duffing_fit(nk, g, g[f"idx_{seed}_0"], 10)
        first.append(harness.validate_dyn_sys(reg, g[f"traj_{seed}"], g[f"ctrl_{seed}"], relative=True))
    dev = np.abs(np.array(first) - g["shipped_first_col"]) / g["shipped_first_col"]
    print("first column vs the shipped all_rmses_nystrom_double_dataset.csv, seeds 0..7:", dev)
    assert dev.max() < 1e-5, dev


@pytest.mark.parametrize("m", [10, 48, 200])
def test_duffing_full_shape_operators_vs_reference(nk, golden, m):
    """Operators of (seed 0, m) at n = 69 900 against the reference's; bar = max(10 x the movement of the reference's own
    operators under the 1e-15 input perturbation or a change of row order, 3 x their movement when the reference's two
    solves go through another LAPACK driver (make_golden_envelope.py), 1e-9)."""
    g = golden("f12_duffing_full.npz")
    e = golden("f12b_duffing_envelope.npz")
    k = int(np.where(g["ms"] == m)[0][0])
    reg = _duffing_fit(nk, g, g[f"idx_0_{k}"], m)
    op_env = float(e["op_envelope"][int(np.where(e["op_envelope_m"] == m)[0][0])])
    op_row = float(e["op_roworder"][int(np.where(e["op_envelope_m"] == m)[0][0])])
    bar = max(10.0 * float(g["op_sensitivity"][0, k]), 10.0 * op_row, 3.0 * op_env, 1e-9)
    errs = dict(A=relf(reg.A, g[f"A_m{m}"]), B=relf(reg.B, g[f"B_m{m}"]), C=relf(reg.C, g[f"C_m{m}"]))
    print(f"\nduffing n=69900 m={m}: {errs}, bar {bar:.2e} (reference moves {float(g['op_sensitivity'][0, k]):.2e})")
    assert max(errs.values()) <= bar, (errs, bar)


def test_duffing_plant_in_the_loop_lqr_vs_reference(nk, O, golden):
    """benchmark_lqr_classic.py:256-299: m = 20 fit on the full data set, K = dlqr(A, B, C^T C, I), 2000 steps of u = K (phi(0)
    - phi(x)) with the PLANT in the loop (a lift per step) and the open-loop replay of the controls (:91-97), seeds 0..2,
    against the reference's own run (plant = the oracle's restatement of dynamical_systems.py:25-48; test infrastructure).
    Bars, per seed and per quantity: max(10 x what the reference's own chain (fit -> DARE -> 2000 feedback steps) moves by
    when its inputs are perturbed by 1e-15 (`lqr_sens`) or its samples arrive in another row order (`lqr_roworder`), 3 x what
    it moves by when its two solves go through another LAPACK driver (`lqr_envelope`, make_golden_envelope.py), 1e-8)."""
    from nys_koop_lqr_amd import harness
    from nys_koop_lqr_amd.lqr import dlqr
    g = golden("f12_duffing_full.npz")
    env = golden("f12b_duffing_envelope.npz")
    steps = int(g["lqr_steps"])
    plant = lambda x, u: O.duffing_step(x, u, 0.01)
    x0, ref0 = np.array([[-0.5], [0.0]]), np.zeros((2, 1))
    for seed in g["seeds"]:
        reg = _duffing_fit(nk, g, g[f"lqr_idx_{seed}"], 20)
        e_ops = max(relf(reg.A, g[f"lqr_A_{seed}"]), relf(reg.B, g[f"lqr_B_{seed}"]), relf(reg.C, g[f"lqr_C_{seed}"]))
        K, _ = dlqr(reg.A, reg.B, reg.C.T @ reg.C, np.eye(1))
        e_K = relf(K, g[f"lqr_K_{seed}"])
        t0 = time.time()
        _, us = harness.lqr_control_plant(steps, ref0, x0, reg, K, plant)
        dt = time.time() - t0
        states = harness.open_loop_control(plant, x0, us)
        e_u, e_x = relf(us, g[f"lqr_us_{seed}"]), relf(states, g[f"lqr_states_{seed}"])
        print(f"\nduffing LQR seed {seed}: operators {e_ops:.2e}, K {e_K:.2e}, controls {e_u:.2e}, states {e_x:.2e} "
              f"({steps} plant-in-the-loop steps in {dt:.2f} s)")
        bars = np.maximum(np.maximum(10.0 * np.maximum(g[f"lqr_sens_{seed}"], env[f"lqr_roworder_{seed}"]),
                                     3.0 * env[f"lqr_envelope_{seed}"]), 1e-8)
        print("   bars:", bars)
        assert e_ops <= bars[0] and e_K <= bars[1] and e_u <= bars[2] and e_x <= bars[3]
