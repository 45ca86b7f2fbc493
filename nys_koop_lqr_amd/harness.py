"""Counterparts of the callers either side of the fit in the reference's benchmark drivers
(benchmark_lqr_cloth.py / _classic.py / _hjb.py): open-loop validation, the (kernel, gamma, m) x K-fold
hyper-parameter sweep, the lifted closed loop, and the cloth data-matrix assembly.  Heavy loops run on the device
through the estimator's C-ABI calls; bookkeeping stays in Python.
"""
import itertools

import numpy as np

from . import _lib
from .regressors import KoopmanNystromRegressor


def open_loop_forecast(regressor, true_trajectory, test_controls):
    """The simulated trajectory of validate_dyn_sys (benchmark_lqr_cloth.py:23-32): lift the first state, then
    z <- A z + B u_i for i < T-1, x = C z; T = number of columns of the true trajectory.  Only the first T-1 control
    columns are used, so `test_controls` may have T-1 columns (benchmark_lqr_hjb.py:129-139 builds them that way) or more."""
    T = true_trajectory.shape[1]
    U = np.asarray(test_controls, dtype=np.float64)[:, :T]
    if U.shape[1] < T - 1:
        raise ValueError(f"{T - 1} control columns needed, got {U.shape[1]}")
    if U.shape[1] == T - 1:  # the rollout call takes one column per state; the last one is never read
        U = np.hstack((U, np.zeros((U.shape[0], 1))))
    return regressor.rollout(true_trajectory[:, 0], U)


def validate_dyn_sys(regressor, true_trajectory, test_controls, relative=False):
    """benchmark_lqr_cloth.py:18-36 (absolute RMSE, :34); relative=True gives the %-form of
    benchmark_lqr_classic.py:39 / benchmark_lqr_hjb.py:42."""
    sim = open_loop_forecast(regressor, true_trajectory, test_controls)
    if relative:
        return np.sqrt(np.sum(np.square(true_trajectory - sim))) / np.sqrt(np.sum(np.square(sim))) * 100
    return np.sqrt(np.mean(np.square(true_trajectory - sim)))


def validate_dyn_sys_all(regressor, true_trajectories, test_controls, relative=False):
    """validate_dyn_sys for ALL test trajectories in one device call (the reference loops over them,
    benchmark_lqr_cloth.py:171-176, benchmark_lqr_hjb.py:296-305): trajectories (k, d, T) with controls (k, p, T) or
    (k, p, T-1) -> k errors.  The trajectories of a batch do not influence each other (same bits as one call each)."""
    trajs = np.asarray(true_trajectories, dtype=np.float64)
    ctrl = np.asarray(test_controls, dtype=np.float64)
    k, d, T = trajs.shape
    if ctrl.shape[0] != k or ctrl.shape[2] < T - 1:
        raise ValueError(f"controls have shape {ctrl.shape}, expected ({k}, p, >= {T - 1})")
    U = np.zeros((k, T, ctrl.shape[1]))
    U[:, : min(T, ctrl.shape[2]), :] = np.transpose(ctrl[:, :, :T], (0, 2, 1))
    if not hasattr(regressor, "_ensure_model"):  # an estimator without a device model (the exact-kernel comparator)
        return np.array([validate_dyn_sys(regressor, trajs[i], ctrl[i], relative) for i in range(k)])
    sims = np.transpose(regressor.rollout(np.ascontiguousarray(trajs[:, :, 0]), U), (0, 2, 1))  # (k, d, T)
    if relative:
        return np.sqrt(np.sum(np.square(trajs - sims), axis=(1, 2))) / np.sqrt(np.sum(np.square(sims), axis=(1, 2))) * 100
    return np.sqrt(np.mean(np.square(trajs - sims), axis=(1, 2)))


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


def cv_unit_score(X, Y, n_inputs, params, fold, centers_idx=None, error_score=np.nan):
    """One (candidate, fold) unit: fit on the training rows (two contiguous ranges, no copy), score the held-out
    rows with sklearn's 'neg_root_mean_squared_error' reduced on the device.  A fit that fails numerically
    (LinAlgError: only possible in strict mode, or when the square-root iteration diverges) scores `error_score`,
    like GridSearchCV's default error_score=nan; error_score='raise' re-raises."""
    n = X.shape[0]
    lo, hi = fold
    n_train = n - (hi - lo)
    reg = KoopmanNystromRegressor(n_inputs, **params)
    if centers_idx is None:  # what clone()+fit does in the reference: fresh draw from the global legacy RNG
        centers_idx = np.random.choice(np.arange(0, n_train), size=reg.m, replace=False)
    centers_idx = np.asarray(centers_idx)
    rows = np.where(centers_idx < lo, centers_idx, centers_idx + (hi - lo))  # training-row index -> dataset row
    reg.nystrom_centers_output = np.asarray(Y)[rows].T
    try:
        reg.fit(X, Y, row_ranges=[(0, lo), (hi, n)], fetch=False)  # the sweep only scores: A, B, C stay on the device
        return reg.score_neg_rmse(X[lo:hi], Y[lo:hi])
    except (np.linalg.LinAlgError, _lib.NyskoopError) as e:
        if isinstance(error_score, str) and error_score == "raise":
            raise
        if isinstance(e, _lib.NyskoopError) and e.code != -5:  # only numerical failures are scored; the rest is a bug
            raise
        return float(error_score)


def _rank_candidates(scores):
    """mean_test_score, best_index as GridSearchCV computes them: a candidate with a failed (NaN) fold has a NaN mean and
    ranks last; best = first candidate with the highest finite mean (-1 if every candidate failed)."""
    mean = scores.mean(axis=1)
    finite = np.isfinite(mean)
    best = int(np.argmax(np.where(finite, mean, -np.inf))) if finite.any() else -1
    return mean, best


def grid_search_cv(X, Y, n_inputs, candidates, n_splits=5, centers=None, work=None, workers=1, error_score=np.nan,
                   batch=0, batch_groups=1):
    """learn_hyperparams (benchmark_lqr_cloth.py:39-66 and the classic/hjb twins) without sklearn's process pool.

    candidates: list of dicts with keys kernel / gamma / m.  centers: optional {(c, f): landmark indices into the
    training rows}; otherwise indices are drawn from the global NumPy RNG in GridSearchCV's order, which reproduces
    sklearn with n_jobs=1 exactly.  work: optional subset of (candidate, fold) units (for sharding over GPUs); units
    not evaluated are NaN.  workers: host threads issuing units concurrently on this GPU (each thread has its own
    context and streams; small fits are latency bound, so several in flight fill the chip).  The landmark draws happen
    up front in GridSearchCV's order, so the scores do not depend on `workers`.  error_score: score of a unit whose fit
    fails numerically (GridSearchCV's default: nan; such a candidate ranks last), or 'raise'.  batch: run that many
    units in lock step (include/nyskoop.h, nk_group_create): small fits are chains of launch-bound kernels, a batch shares
    every launch; scores are bit-identical to batch=0.
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
            idx = centers.get((c, f)) if (c, f) not in mine else centers[(c, f)]
        else:  # drawn for every unit, evaluated or not: keeps the RNG stream aligned with the serial sweep
            n_train = X.shape[0] - (folds[f][1] - folds[f][0])
            idx = np.random.choice(np.arange(0, n_train), size=candidates[c]["m"], replace=False)
        if (c, f) in mine:
            todo.append((c, f, idx))

    def run(item):
        c, f, idx = item
        return c, f, cv_unit_score(X, Y, n_inputs, candidates[c], folds[f], idx, error_score)

    if batch > 1 and len(todo) > 1:
        # lock-step batching (nk_cv_grid): `batch` units per round, one library thread each, kernel launches merged
        units = []
        for (c, f, idx) in todo:
            lo, hi = folds[f]
            idx = np.asarray(idx)
            rows = np.where(idx < lo, idx, idx + (hi - lo))  # training-row index -> dataset row
            kern = candidates[c]["kernel"].kernel
            units.append((kern, candidates[c]["gamma"], 1e-6, candidates[c]["m"], (lo, hi), rows))
        if batch_groups > 1 and len(units) >= 2 * batch:
            # several independent lock-step groups, each on its own stream and driven from its own host thread: the
            # latency-bound factorisation chains of one group overlap with the GEMM-bound stages of another
            from concurrent.futures import ThreadPoolExecutor
            shares = [list(range(gi, len(units), batch_groups)) for gi in range(batch_groups)]
            sc, status = np.full(len(units), np.nan), np.zeros(len(units), dtype=np.int32)

            def run_share(gi):
                sub = [units[i] for i in shares[gi]]
                return _lib.lockstep_pool(batch, index=gi).cv_grid(X, Y, n_inputs, sub)

            with ThreadPoolExecutor(max_workers=batch_groups) as ex:
                for gi, (s_g, st_g) in enumerate(ex.map(run_share, range(batch_groups))):
                    sc[shares[gi]], status[shares[gi]] = s_g, st_g
        else:
            sc, status = _lib.lockstep_pool(batch).cv_grid(X, Y, n_inputs, units)
        bad = [int(st) for st in status if st not in (0, -3, -5)]
        if bad:
            raise _lib.NyskoopError(bad[0], "a unit of the batched sweep failed")
        if isinstance(error_score, str) and error_score == "raise" and np.any(status != 0):
            raise np.linalg.LinAlgError("a unit of the batched sweep failed numerically")
        sc = np.where(status == 0, sc, float("nan") if isinstance(error_score, str) else float(error_score))
        results = [(c, f, float(s)) for (c, f, _), s in zip(todo, sc)]
    elif workers > 1 and len(todo) > 1:
        results = list(_lib.worker_pool(workers).map(run, todo))
    else:
        results = [run(item) for item in todo]
    for c, f, sc in results:
        scores[c, f] = sc
    if work is None:
        mean, best = _rank_candidates(scores)
    else:  # a shard: the caller assembles the full table
        mean, best = scores.mean(axis=1), -1
    return dict(split_scores=scores, mean_test_score=mean, best_index=best,
                best_params=candidates[best] if best >= 0 else None)


def lqr_closed_loop(num_steps, reference, initial_state, regressor, K):
    """The loop of benchmark_lqr_cloth.py:73-84 alone: returns (visited_states (d, 1+num_steps) starting with the initial
    state, u_ops (p, num_steps)).  One device call (lifts of both states + the whole lifted recursion)."""
    initial_state = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    reference = np.asarray(reference, dtype=np.float64).reshape(-1, 1)
    phi = regressor.lift(np.hstack((initial_state, reference)))  # one call for both lifts
    xs, us = regressor.closed_loop(K, phi[:, 0], phi[:, 1], num_steps)
    return np.hstack((initial_state, xs)), us


def lqr_control(num_steps, reference, initial_state, regressor, K, control_nodes=(168, 169, 170, 189, 190, 191),
                simulator_order=(0, 3, 1, 4, 2, 5)):
    """benchmark_lqr_cloth.py:69-104 in full: the lifted closed loop, the CUMULATIVE input sequence seeded with the
    positions of the two controlled corner nodes (`u_s[:, 0] = initial_state[control_nodes]`, every later column adds
    the step's u_op, :76-81), the per-axis split of the visited states (x = rows 0,3,6.., y = 1,4,.., z = 2,5,..; :85-101)
    and the input rows permuted for the MATLAB simulator (:102).  Returns (x_s, y_s, z_s, final_us) with
    x_s, y_s, z_s: (d/3, 1+num_steps), final_us: (p, 1+num_steps)."""
    initial_state = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    visited, u_ops = lqr_closed_loop(num_steps, reference, initial_state, regressor, K)
    u0 = initial_state[list(control_nodes), :]
    u_s = np.hstack((u0, u0 + np.cumsum(u_ops, axis=1)))
    x_s, y_s, z_s = visited[0::3], visited[1::3], visited[2::3]
    final_us = u_s[list(simulator_order), :]
    return x_s, y_s, z_s, final_us


def lqr_control_plant(num_steps, reference, initial_state, regressor, K, plant_step):
    """The plant-in-the-loop variant of benchmark_lqr_hjb.py:73-97 (and _classic.py:67-89): u = K (phi(ref) - phi(x)),
    x <- plant_step(x, u), phi re-lifted from the true state every step (one cached-square-root lift per step instead of
    the reference's O(m^3) sqrtm).  plant_step(x (d,1), u (p,1)) -> x_next.  Returns (x_s (num_steps,), u_s (p, num_steps))
    like the reference: x_s is the first state coordinate of the visited states."""
    x = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    phi_ref = regressor.lift(np.asarray(reference, dtype=np.float64).reshape(-1, 1))
    phi = regressor.lift(x)
    xs, us = [], []
    for _ in range(num_steps):
        u = K @ (phi_ref - phi)
        us.append(u.reshape(-1, 1))
        xs.append(x[0, 0])
        x = np.asarray(plant_step(x, u), dtype=np.float64).reshape(-1, 1)
        phi = regressor.lift(x)
    return np.array(xs), np.hstack(us)


def open_loop_control(plant_step, initial_state, controls):
    """benchmark_lqr_classic.py:91-97: replay a control sequence (p x T) on the true plant; returns the visited states
    (d x (T + 1)), the initial state first."""
    state = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    controls = np.asarray(controls, dtype=np.float64)
    states = [state]
    for i in range(controls.shape[1]):
        state = np.asarray(plant_step(state, controls[:, i].reshape(-1, 1)), dtype=np.float64).reshape(-1, 1)
        states.append(state)
    return np.hstack(states)


def control_rmse_percent(us, u_opt):
    """benchmark_lqr_hjb.py:313 / :378."""
    us, u_opt = np.asarray(us).squeeze(), np.asarray(u_opt).squeeze()
    return np.sqrt(np.sum(np.square(us - u_opt))) / np.sqrt(np.sum(np.square(u_opt))) * 100


def create_data_matrices(trajs, controls, indices):
    """benchmark_lqr_cloth.py:117-130: snapshot pairs from trajectories (d x T) and controls (p x T)."""
    states = np.hstack([trajs[i][:, :-1] for i in indices])
    next_states = np.hstack([trajs[i][:, 1:] for i in indices])
    inputs = np.hstack([controls[i][:, :-1] for i in indices])
    return np.vstack((states, inputs)), next_states
