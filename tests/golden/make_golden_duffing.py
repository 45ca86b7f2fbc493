#!/usr/bin/env python3
"""Config 1 at the reference driver's REAL shape (benchmark_lqr_classic.py:174-179,211-255): golden vectors produced by
RUNNING THE REFERENCE in the build container.  Only DATA is written: the reference's data set, seeds, landmark indices,
the reference's outputs and three rows of a result file the authors shipped.

    python tests/golden/make_golden_duffing.py

  f12_duffing_full.npz
    * X (69 900 x 3), Y (69 900 x 2): duffing/duffing_{x,y,u}_{forced,unforced}.csv assembled as :174-178 does;
    * open-loop validation of :211-255 for seeds 0..2: np.random.seed(seed) -> test trajectory (simulate_true_system,
      :124-135) ; np.random.seed(seed) -> 20 fits with m = around(logspace(1, 2.3, 20)) (:179), KernelWrapper([1, 1]),
      gamma = 1e-6, landmarks drawn by the reference from the global legacy RNG (regressors.py:130) -> relative-% RMSE
      (:39).  The hyper-parameters sit in a pickled GridSearchCV (not loaded: pickle executes code); gamma = 1e-6 is
      the ONE value of the script's grid (:50) for which the reference reproduces the FIRST COLUMN (m = 10, the first
      fit after np.random.seed(seed)) of the shipped duffing/all_rmses_nystrom_double_dataset.csv: to 1e-8..1e-11 for
      every seed tried (0..7 are stored: `shipped_first_col`, with the landmarks and test trajectories of seeds 3..7).
      The later columns of that file do not reproduce (10 %..5 x off, whatever m schedule / reseeding / interleaved
      spline fits were tried): the run that wrote it consumed the RNG differently between fits, so for m > 10 the
      golden values are the reference's own outputs here (`ref_rmse`), not the authors' file;
    * the same 60 fits with X perturbed by one part in 1e15: the reference's own reproducibility, fit by fit (the
      parity bar of the GPU test is a fixed multiple of it);
    * operators A, B, C of (seed 0; m = 10, 48, 200) for operator-level parity;
    * plant-in-the-loop LQR of :67-89,256-299 for seeds 0..2: m = 20, K = dlqr(A, B, C^T C, I), 2000 steps with a lift per
      step, then the open-loop replay of the controls (:91-97).  (python-control is absent: K from SciPy's DARE, as in
      make_golden_configs.py.)
"""
import os
import random
import sys
import time

import numpy as np
import scipy.linalg
import scipy.signal

REF = os.environ.get("NK_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import regressors as R  # noqa: E402  (the reference)
import dynamical_systems as DS  # noqa: E402

DUF = f"{REF}/duffing"


def load_dataset():
    """benchmark_lqr_classic.py:174-178."""
    ld = lambda f: np.loadtxt(f"{DUF}/{f}", delimiter=",")
    xf, xu = ld("duffing_x_forced.csv"), ld("duffing_x_unforced.csv")
    X = np.hstack((xf, xu))
    U = np.hstack((ld("duffing_u_forced.csv").reshape([1, -1]), np.zeros((1, xu.shape[1]))))
    X = np.vstack((X, U))
    Y = np.hstack((ld("duffing_y_forced.csv"), ld("duffing_y_unforced.csv")))
    return X, Y  # (3, n), (2, n)


def simulate_true_system(ds, T):
    """:124-135."""
    length = np.sqrt(np.random.uniform(0, ds.radius_sampling))
    angle = np.pi * np.random.uniform(0, ds.angle_sampling)
    state = np.array([length * np.cos(angle), length * np.sin(angle)]).reshape([-1, 1])
    times = np.linspace(0, T, int(1 / ds.Ts))
    u_s = 1.0 * scipy.signal.square(2 * np.pi * 10 / 3 * times)
    visited = state.reshape([-1, 1])
    for u in u_s:
        state = ds.update_SOM(state, u)
        visited = np.hstack((visited, state.reshape([-1, 1])))
    return visited, u_s.reshape([ds.n_inputs, -1])


def validate_dyn_sys(reg, traj, controls):
    """:23-41."""
    x = reg.lift(traj[:, 0].reshape([-1, 1]))
    sim = reg.C @ x
    for i in range(traj.shape[1] - 1):
        x = reg.A @ x + reg.B @ controls[:, i].reshape([-1, 1])
        sim = np.hstack((sim, reg.C @ x))
    return np.sqrt(np.sum(np.square(traj - sim))) / np.sqrt(np.sum(np.square(sim))) * 100, sim


def dlqr(A, B, Q, R_):
    P = scipy.linalg.solve_discrete_are(A, B, Q, R_)
    return np.linalg.solve(B.T @ P @ B + R_, B.T @ P @ A)


def closed_loop(ds, reg, x0, reference, steps):
    """benchmark_lqr_classic.py:256-299 for one fitted regressor: K = dlqr(A, B, C^T C, I), lqr_control (:67-89, plant in the
    loop, a lift per step), open_loop_control (:91-97)."""
    K = dlqr(reg.A, reg.B, reg.C.T @ reg.C, np.eye(1))
    phi_new, phi_ref = reg.lift(x0), reg.lift(reference)
    visited, u_s, x_new = x0, np.empty((1, 0)), x0
    for _ in range(steps):
        u_op = K @ (phi_ref - phi_new)
        u_s = np.hstack((u_s, u_op.reshape(1, 1)))
        visited = np.hstack((visited, reg.C @ phi_new))
        x_new = ds.update_SOM(x_new, u_op)
        phi_new = reg.lift(x_new)
    state, states = x0, x0.reshape([-1, 1])
    for i in range(u_s.shape[1]):
        state = ds.update_SOM(state, u_s[:, i])
        states = np.hstack((states, state))
    return K, visited, u_s, states


def duffing_plant():
    return DS.DuffingOscillator(Ts=0.01, name="duffing", n_states=2, n_inputs=1, radius_sampling=1.0, angle_sampling=2,
                                input_lb=[-1], input_ub=[1])


class ChoiceSpy:
    """Observes np.random.choice (regressors.py:130) to record the landmark indices the reference drew."""

    def __init__(self):
        self.draws = []
        self._orig = np.random.choice

    def __enter__(self):
        def spy(*a, **kw):
            out = self._orig(*a, **kw)
            self.draws.append(np.array(out))
            return out
        np.random.choice = spy
        return self

    def __exit__(self, *exc):
        np.random.choice = self._orig


def main():
    ds = DS.DuffingOscillator(Ts=0.01, name="duffing", n_states=2, n_inputs=1, radius_sampling=1.0, angle_sampling=2,
                              input_lb=[-1], input_ub=[1])
    X, Y = load_dataset()
    n = X.shape[1]
    ms = np.around(np.logspace(1, 2.3, num=20)).astype(int)
    gamma = 1e-6
    shipped = np.loadtxt(f"{DUF}/all_rmses_nystrom_double_dataset.csv")
    seeds = [0, 1, 2]
    out = dict(X=X.T.copy(), Y=Y.T.copy(), ms=ms, gamma=gamma, seeds=np.array(seeds))
    prng = np.random.default_rng(7)
    Xp = X * (1 + 1e-15 * prng.standard_normal(X.shape))
    relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    rm = np.zeros((3, 20)); rm_p = np.zeros((3, 20)); sens_ops = np.zeros((3, 20)); sens_sim = np.zeros((3, 20))
    t0 = time.time()
    for si, seed in enumerate(seeds):
        np.random.seed(seed); random.seed(seed)
        traj, ctrl = simulate_true_system(ds, 2)
        out[f"traj_{seed}"], out[f"ctrl_{seed}"] = traj, ctrl
        np.random.seed(seed); random.seed(seed)
        with ChoiceSpy() as spy:
            for k, m in enumerate(ms):
                reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=gamma, m=int(m))
                reg.fit(X.T, Y.T)
                rm[si, k], sim = validate_dyn_sys(reg, traj, ctrl)
                idx = spy.draws[-1]
                out[f"idx_{seed}_{k}"] = idx
                # the reference's own reproducibility: same landmarks, inputs perturbed by 1e-15
                reg2 = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=gamma, m=int(m))
                reg2.nystrom_centers_output = Y[:, idx]   # landmarks of the UNPERTURBED run (the RNG is not touched)
                reg2.fit(Xp.T, Y.T)
                rm_p[si, k], sim2 = validate_dyn_sys(reg2, traj, ctrl)
                sens_ops[si, k] = max(relf(reg2.A, reg.A), relf(reg2.B, reg.B), relf(reg2.C, reg.C))
                sens_sim[si, k] = relf(sim2, sim)
                if seed == 0 and int(m) in (10, 48, 200):
                    out[f"A_m{m}"], out[f"B_m{m}"], out[f"C_m{m}"], out[f"sim_m{m}"] = reg.A, reg.B, reg.C, sim
        print(f"seed {seed}: rmse moved by the 1e-15 perturbation: max rel {np.max(np.abs(rm_p[si] - rm[si]) / rm[si]):.2e} "
              f"({time.time() - t0:.0f} s)", flush=True)
    out.update(ref_rmse=rm, ref_rmse_perturbed=rm_p, op_sensitivity=sens_ops, sim_sensitivity=sens_sim)
    # first column of the shipped file, seeds 3..7 (one m = 10 fit each)
    first = list(rm[:, 0])
    for seed in range(3, 8):
        np.random.seed(seed); random.seed(seed)
        traj, ctrl = simulate_true_system(ds, 2)
        out[f"traj_{seed}"], out[f"ctrl_{seed}"] = traj, ctrl
        np.random.seed(seed); random.seed(seed)
        with ChoiceSpy() as spy:
            reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=gamma, m=10)
            reg.fit(X.T, Y.T)
        out[f"idx_{seed}_0"] = spy.draws[-1]
        first.append(validate_dyn_sys(reg, traj, ctrl)[0])
    first = np.array(first)
    print("first column vs the shipped file, seeds 0..7: rel dev", np.abs(first - shipped[:8, 0]) / shipped[:8, 0])
    assert np.all(np.abs(first - shipped[:8, 0]) / shipped[:8, 0] < 1e-6)
    out.update(shipped_first_col=shipped[:8, 0].copy(), ref_first_col=first)
    # ---- plant-in-the-loop LQR (:256-299): m = 20 ----------------------------------------------------------------
    steps = int(10 * 2 / ds.Ts)
    x0 = np.array([-0.5, 0.0]).reshape([-1, 1])
    reference = np.array([0.0, 0.0]).reshape([-1, 1])
    for seed in seeds:
        np.random.seed(seed); random.seed(seed)
        with ChoiceSpy() as spy:
            reg = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=gamma, m=20)
            reg.fit(X.T, Y.T)
        K, visited, u_s, states = closed_loop(ds, reg, x0, reference, steps)
        # the reference's own reproducibility of the whole chain (fit -> DARE -> 2000 feedback steps): same landmarks,
        # inputs perturbed by 1e-15
        reg2 = R.KoopmanNystromRegressor(1, kernel=R.KernelWrapper([1, 1]), gamma=gamma, m=20)
        reg2.nystrom_centers_output = Y[:, spy.draws[-1]]
        reg2.fit(Xp.T, Y.T)
        K2, _, u2, st2 = closed_loop(ds, reg2, x0, reference, steps)
        out[f"lqr_sens_{seed}"] = np.array([max(relf(reg2.A, reg.A), relf(reg2.B, reg.B), relf(reg2.C, reg.C)),
                                            relf(K2, K), relf(u2, u_s), relf(st2, states)])
        out[f"lqr_idx_{seed}"], out[f"lqr_K_{seed}"] = spy.draws[-1], K
        out[f"lqr_A_{seed}"], out[f"lqr_B_{seed}"], out[f"lqr_C_{seed}"] = reg.A, reg.B, reg.C
        out[f"lqr_visited_{seed}"], out[f"lqr_us_{seed}"], out[f"lqr_states_{seed}"] = visited, u_s, states
        print(f"lqr seed {seed}: final state {states[:, -1]}, |u| max {np.abs(u_s).max():.3f}; the 1e-15 perturbation moves "
              f"(operators, K, controls, states) by {out[f'lqr_sens_{seed}']}", flush=True)
    out["lqr_steps"] = steps
    np.savez_compressed(f"{OUT}/f12_duffing_full.npz", **out)
    print("wrote f12_duffing_full.npz", os.path.getsize(f"{OUT}/f12_duffing_full.npz") / 1e6, "MB")


if __name__ == "__main__":
    main()
