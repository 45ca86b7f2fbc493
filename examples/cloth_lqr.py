#!/usr/bin/env python3
"""The flow of the reference's benchmark_lqr_cloth.py (:140-270) on this library, with the reference's own data from the committed
fixtures: data matrices from trajectories -> hyper-parameter search (a 2 x 2 corner of the 27 x 3 grid) -> fit on the thirty
training trajectories -> open-loop validation -> LQR gain (host DARE) against the authors' shipped K_lqr_seed_0.csv -> closed loop
in the lifted space.  Needs an MI355X (the library has no CPU path):

    python3 examples/cloth_lqr.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import harness
from nys_koop_lqr_amd.lqr import cloth_gain_for_simulator

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
t = np.load(os.path.join(G, "cloth_trajs_all.npz"))
states, inputs = t["states_e10"] / 1e10, t["inputs"]          # 50 trajectories: (192 x T), (6 x T)
known = np.load(os.path.join(G, "f6_cloth_known_gain.npz"))

def data_matrices(ids):                                        # benchmark_lqr_cloth.py:117-130
    X = np.hstack([np.vstack((states[i][:, :-1], inputs[i][:, :-1])) for i in ids]).T
    Y = np.hstack([states[i][:, 1:] for i in ids]).T
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)

# ---- learn_hyperparams (:39-66) on the ten validation trajectories: (kernel, gamma) x 5 folds, lock-step batched
Xv, Yv = data_matrices(range(10))
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, 192), gamma=g, m=500) for l in (10.0, 100.0) for g in (1e-6, 1e-5)]
np.random.seed(42)
t0 = time.perf_counter()
cv = harness.grid_search_cv(Xv, Yv, 6, cands, n_splits=5, batch=20, batch_groups=1)
print(f"search: {len(cands) * 5} units in {time.perf_counter() - t0:.2f} s; mean scores {np.round(cv['mean_test_score'], 5)}; "
      f"best candidate {cv['best_index']}")

# ---- the run the authors shipped a gain for (:218-263): seed 0, m = 100, l = 10, gamma = 1e-7, trajectories 10..39
X, Y = data_matrices(range(10, 40))
np.random.seed(0)
reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(10, 10, 10, 192), gamma=1e-7, m=100)
t0 = time.perf_counter()
reg.fit(X, Y)
print(f"fit: n = {X.shape[0]}, m = 100 in {1e3 * (time.perf_counter() - t0):.1f} ms; A {reg.A.shape} B {reg.B.shape} C {reg.C.shape}")
rmse = [harness.validate_dyn_sys(reg, states[i], inputs[i]) for i in range(40, 50)]
print(f"open-loop RMSE on the ten test trajectories (:18-36): median {np.median(rmse):.3e}")
K = reg.solve_lqr(c=0.005)                                     # Q = c C'C, R = I, scipy DARE (:238-263)
Ks = cloth_gain_for_simulator(K)                               # rows permuted for the MATLAB simulator (:263)
rel = np.linalg.norm(Ks - known["K_lqr_seed_0"]) / np.linalg.norm(known["K_lqr_seed_0"])
print(f"LQR gain {K.shape}: {rel:.3e} from the authors' K_lqr_seed_0.csv (the reference with LAPACK's other drivers: 2.4e-2)")
phi0, phir = reg.lift(states[10][:, :1]), reg.lift(states[10][:, 50:51])
xs, us = reg.closed_loop(K, phi0, phir, 60)                    # lqr_control in the lifted space (:69-104), one launch
print(f"closed loop: 60 steps in one launch, distance to the reference state {np.linalg.norm(states[10][:, 0] - states[10][:, 50]):.3f} -> "
      f"{np.linalg.norm(xs[:, -1] - states[10][:, 50]):.3f}; largest control {np.abs(us).max():.3f}")
