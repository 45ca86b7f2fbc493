#!/usr/bin/env python3
"""Per-unit table of the real cloth CV grid (405 units, tests/golden/f7_cloth_cv_full.npz): what the reference's gelsd did
(rank, sigma_min / sigma_max of both regularised systems), what the build did (rank used, Cholesky or rank-truncating
branch), the score error against the reference and the two reference-side bars (f7b: reproducibility spread, LAPACK
envelope).  python3 tools/cloth_grid_units.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(f"{G}/f7_cloth_cv_full.npz"); t = np.load(f"{G}/cloth_trajs_all.npz")
env = np.load(f"{G}/f7b_cloth_cv_envelope.npz") if os.path.exists(f"{G}/f7b_cloth_cv_envelope.npz") else None
st = t["states_e10"] / 1e10
X = np.ascontiguousarray(np.hstack([np.vstack((st[i][:, :-1], t["inputs"][i][:, :-1])) for i in range(10)]).T)
Y = np.ascontiguousarray(np.hstack([st[i][:, 1:] for i in range(10)]).T)
folds = harness.kfold_slices(1010, 5)
np.random.seed(int(g["seed"]))
rows = []
t0 = time.perf_counter()
for c in range(81):
    ls = g["ls_grid"][int(g["order_kernel"][c])]; gam = float(g["order_gamma"][c])
    for f, (lo, hi) in enumerate(folds):
        idx = np.random.choice(np.arange(0, 1010 - (hi - lo)), size=500, replace=False)
        r = np.where(idx < lo, idx, idx + (hi - lo))
        reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(*ls, 192), gamma=gam, m=500)
        reg.nystrom_centers_output = Y[r].T
        reg.fit(X, Y, row_ranges=[(0, lo), (hi, 1010)], fetch=False)
        sc = reg.score_neg_rmse(X[lo:hi], Y[lo:hi])
        s = reg.fit_stats_
        ref = g["split_scores"][c, f]
        err_svd, rank_svd = float("nan"), (0, 0)
        if gam < 5e-7:  # the same unit in the lstsq-shaped mode (both systems through the SVD, gelsd's cut-off)
            nk.get_context().set_strict_spd(2)
            try:
                reg2 = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(*ls, 192), gamma=gam, m=500)
                reg2.nystrom_centers_output = Y[r].T
                reg2.fit(X, Y, row_ranges=[(0, lo), (hi, 1010)], fetch=False)
                err_svd = abs(reg2.score_neg_rmse(X[lo:hi], Y[lo:hi]) - ref) / abs(ref)
                rank_svd = (reg2.fit_stats_["rank_inner"], reg2.fit_stats_["rank_inner_rec"])
            finally:
                nk.get_context().set_strict_spd(0)
        rows.append((c, f, tuple(int(v) for v in ls), gam, int(g["lstsq_rank"][c, f, 0]), int(g["lstsq_rank"][c, f, 1]),
                     g["lstsq_smin"][c, f, 0] / g["lstsq_smax"][c, f, 0], s["rank_inner"], s["rank_inner_rec"],
                     abs(sc - ref) / abs(ref), env["spread"][c, f] if env else np.nan, env["envelope"][c, f] if env else np.nan,
                     err_svd, rank_svd))
dt = time.perf_counter() - t0
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
print(f"# 405 units one at a time in {dt:.2f} s.  gelsd rank: inner (of 506) / inner_rec (of 500); build rank: 506 / 500 = Cholesky at full rank", file=out)
print("# last two columns (gamma = 1e-7 only): the same unit with nk_set_strict_spd(ctx, 2) -- both systems through the SVD with gelsd's cut-off: error, ranks", file=out)
print("cand fold  ls              gamma  gelsd_rank   sv_ratio  build_rank  rel.err   spread    envelope  err/bar  err_lstsq_mode  ranks", file=out)
for r in rows:
    bar = max(10 * r[10], 1.5 * r[11], 1e-7) if env else np.nan
    print("%4d %4d  %-14s %6.0e  %4d/%-4d  %9.2e  %4d/%-4d  %8.2e  %8.2e  %8.2e  %6.2f  %8.2e  %s" %
          (r[0], r[1], str(r[2]), r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[9] / bar, r[12],
           ("%d/%d" % r[13]) if r[13][0] else "-"), file=out)
err = np.array([r[9] for r in rows]); gam = np.array([r[3] for r in rows])
trunc_ref = np.array([r[4] < 506 or r[5] < 500 for r in rows]); trunc_us = np.array([r[7] < 506 or r[8] < 500 for r in rows])
print(f"# reference truncated {trunc_ref.sum()} units, build {trunc_us.sum()} (both {np.sum(trunc_ref & trunc_us)})", file=out)
e7 = np.isclose(gam, 1e-7, rtol=1e-6)
esvd = np.array([r[12] for r in rows])
print(f"# gamma 1e-07, default (Cholesky unless a pivot fails) against the lstsq-shaped mode: max {err[e7].max():.2e} / {np.nanmax(esvd[e7]):.2e}, "
      f"median {np.median(err[e7]):.2e} / {np.nanmedian(esvd[e7]):.2e}; units where the lstsq-shaped mode is closer to the reference: "
      f"{int(np.sum(esvd[e7] < err[e7]))} of {int(e7.sum())}", file=out)
for gv in (1e-7, 1e-6, 1e-5):
    sel = np.isclose(gam, gv, rtol=1e-6)
    print(f"# gamma {gv:.0e}: max {err[sel].max():.2e} median {np.median(err[sel]):.2e}; among units the reference truncated: "
          f"max {err[sel & trunc_ref].max() if (sel & trunc_ref).any() else 0:.2e}", file=out)
