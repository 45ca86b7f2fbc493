"""Counterparts of the callers either side of the fit in the reference's benchmark drivers
(benchmark_lqr_cloth.py / _classic.py / _hjb.py): open-loop validation, the (kernel, gamma, m) x K-fold
hyper-parameter sweep, the lifted closed loop, and the cloth data-matrix assembly.  Heavy loops run on the device
through the estimator's C-ABI calls; bookkeeping stays in Python.
"""
import itertools

import numpy as np

from . import _lib
from .regressors import KoopmanNystromRegressor


def validate_dyn_sys(regressor, true_trajectory, test_controls, relative=False):
    """benchmark_lqr_cloth.py:18-36 (absolute RMSE, :34); relative=True gives the %-form of
    benchmark_lqr_classic.py:39 / benchmark_lqr_hjb.py:42."""
    T = true_trajectory.shape[1]
    sim = regressor.rollout(true_trajectory[:, 0], np.asarray(test_controls)[:, :T])
    if relative:
        return np.sqrt(np.sum(np.square(true_trajectory - sim))) / np.sqrt(np.sum(np.square(sim))) * 100
    return np.sqrt(np.mean(np.square(true_trajectory - sim)))


def kfold_slices(n, n_splits=5):
    """sklearn KFold(n_splits) without shuffling (GridSearchCV's default cv): contiguous test folds, the first
    n % n_splits folds one element longer."""
    sizes = np.full(n_splits, n // n_splits, dtype=int)
    sizes[: n % n_splits] += 1
    out, cur = [], 0
    for s in sizes:
        out.append((cur, cur + int(s)))
        cur += int(s)
    return out


def parameter_grid(grid):
    """Candidates in sklearn ParameterGrid order: keys sorted, last key fastest."""
    keys = sorted(grid)
    return [dict(zip(keys, vals)) for vals in itertools.product(*(grid[k] for k in keys))]


def cv_work_list(n_candidates, n_splits):
    """(candidate, fold) units in GridSearchCV's evaluation order (candidate-major)."""
    return [(c, f) for c in range(n_candidates) for f in range(n_splits)]


def cv_unit_score(X, Y, n_inputs, params, fold, centers_idx=None):
    """One (candidate, fold) unit: fit on the training rows (two contiguous ranges, no copy), score the held-out
    rows with sklearn's 'neg_root_mean_squared_error' reduced on the device."""
    n = X.shape[0]
    lo, hi = fold
    n_train = n - (hi - lo)
    reg = KoopmanNystromRegressor(n_inputs, **params)
    if centers_idx is None:  # what clone()+fit does in the reference: fresh draw from the global legacy RNG
        centers_idx = np.random.choice(np.arange(0, n_train), size=reg.m, replace=False)
    centers_idx = np.asarray(centers_idx)
    rows = np.where(centers_idx < lo, centers_idx, centers_idx + (hi - lo))  # training-row index -> dataset row
    reg.nystrom_centers_output = np.asarray(Y)[rows].T
    reg.fit(X, Y, row_ranges=[(0, lo), (hi, n)])
    return reg.score_neg_rmse(X[lo:hi], Y[lo:hi])


def grid_search_cv(X, Y, n_inputs, candidates, n_splits=5, centers=None, work=None, workers=1):
    """learn_hyperparams (benchmark_lqr_cloth.py:39-66 and the classic/hjb twins) without sklearn's process pool.

    candidates: list of dicts with keys kernel / gamma / m.  centers: optional {(c, f): landmark indices into the
    training rows}; otherwise indices are drawn from the global NumPy RNG in GridSearchCV's order, which reproduces
    sklearn with n_jobs=1 exactly.  work: optional subset of (candidate, fold) units (for sharding over GPUs); units
    not evaluated are NaN.  workers: host threads issuing units concurrently on this GPU (each thread has its own
    context and streams; small fits are latency bound, so several in flight fill the chip).  The landmark draws happen
    up front in GridSearchCV's order, so the scores do not depend on `workers`.
    Returns split_scores (n_cand x n_splits), mean_test_score, best_index.
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    folds = kfold_slices(X.shape[0], n_splits)
    scores = np.full((len(candidates), n_splits), np.nan)
    units = cv_work_list(len(candidates), n_splits)
    mine = set(units if work is None else work)
    todo = []
    for (c, f) in units:
        if centers is not None:
            idx = centers[(c, f)]
        else:  # drawn for every unit, evaluated or not: keeps the RNG stream aligned with the serial sweep
            n_train = X.shape[0] - (folds[f][1] - folds[f][0])
            idx = np.random.choice(np.arange(0, n_train), size=candidates[c]["m"], replace=False)
        if (c, f) in mine:
            todo.append((c, f, idx))

    def run(item):
        c, f, idx = item
        return c, f, cv_unit_score(X, Y, n_inputs, candidates[c], folds[f], idx)

    if workers > 1 and len(todo) > 1:
        results = list(_lib.worker_pool(workers).map(run, todo))
    else:
        results = [run(item) for item in todo]
    for c, f, sc in results:
        scores[c, f] = sc
    mean = scores.mean(axis=1)
    best = int(np.nanargmax(mean)) if np.all(np.isfinite(mean)) else -1
    return dict(split_scores=scores, mean_test_score=mean, best_index=best,
                best_params=candidates[best] if best >= 0 else None)


def lqr_control(num_steps, reference, initial_state, regressor, K):
    """Lifted closed loop of benchmark_lqr_cloth.py:69-84: returns (visited_states (d, 1+num_steps) starting with the
    initial state, u_ops (p, num_steps))."""
    phi_new = regressor.lift(initial_state)
    phi_reference = regressor.lift(reference)
    xs, us = regressor.closed_loop(K, phi_new, phi_reference, num_steps)
    return np.hstack((initial_state.reshape(-1, 1), xs)), us


def create_data_matrices(trajs, controls, indices):
    """benchmark_lqr_cloth.py:117-130: snapshot pairs from trajectories (d x T) and controls (p x T)."""
    states = np.hstack([trajs[i][:, :-1] for i in indices])
    next_states = np.hstack([trajs[i][:, 1:] for i in indices])
    inputs = np.hstack([controls[i][:, :-1] for i in indices])
    return np.vstack((states, inputs)), next_states
