#!/usr/bin/env python3
"""Operator errors of the golden fits with the current NYSKOOP_REFINE_PIVOT (run twice: default and =0) -- what one step of
iterative refinement of the two regularised solves buys per fixture."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
FITS = [("f1_cloth_rbf_wellcond.npz", 6), ("f2_synth_rbf_d384.npz", 6), ("f4_hjb_matern.npz", 1), ("f1_cloth_rbf_illcond.npz", 6),
        ("f3_duffing_matern.npz", 1), ("f8_hjb_config2.npz", 1)]
print("NYSKOOP_REFINE_PIVOT =", os.environ.get("NYSKOOP_REFINE_PIVOT", "(default)"))
for name, p in FITS:
    g = np.load("tests/golden/" + name)
    X, Y = g["X"].astype(np.float64), g["Y"].astype(np.float64)
    d = Y.shape[1]
    ls = np.atleast_1d(g["ls"]).astype(np.float64)
    if "matern" in name or "hjb_config2" in name:
        kern = nk.KernelWrapper(ls)
    else:
        l3 = ls if ls.size == 3 else np.repeat(ls, 3)
        kern = nk.ThreeDimensionalKernel(*l3, d)
    reg = nk.KoopmanNystromRegressor(p, kernel=kern, gamma=float(g["gamma"]), m=len(g["idx"]))
    reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, g["idx"]])
    reg.fit(X, Y)
    st = reg.fit_stats_
    errs = {nm: relf(got, g[nm]) for nm, got in (("A", reg.A), ("B", reg.B), ("C", reg.C))}
    print(f"{name:28s} A {errs['A']:.2e} B {errs['B']:.2e} C {errs['C']:.2e} | piv {st.get('pivot_ratio_inner', -1):.2e} "
          f"{st.get('pivot_ratio_inner_rec', -1):.2e} refined {st.get('refined')} first-correction ratio {st.get('refine_ratio_inner', 0):.1e} {st.get('refine_ratio_inner_rec', 0):.1e}")
