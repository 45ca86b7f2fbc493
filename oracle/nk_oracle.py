"""CPU oracle for the Nystrom-Koopman hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference algorithm
(`/root/reference/regressors.py` and the callers in `benchmark_lqr_*.py`).
It is the *checker* for the HIP path: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The product package
(`nys_koop_lqr_amd`) never imports anything from `oracle/`.

Parity pin: every function here is checked in `tests/test_oracle_golden.py`
against golden vectors produced by importing the reference itself in the build
container (`tests/golden/make_golden.py`, committed next to the vectors) and
against the reference's shipped `K_lqr_seed_*.csv` gains.

Two modes:
  * faithful=True  -- the same call sequence as the reference (cdist direct
    differences, scipy sqrtm, solve(assume_a='her'), lstsq/gelsd).  This is the
    timed "reference-faithful" CPU baseline.
  * faithful=False -- algebraically identical "fast CPU" form (eigh-based
    square root computed once, Cholesky solves), reported separately so GPU
    speed-ups are not inflated by the reference's avoidable O(m^3) work.

Third-party arithmetic restated here (not under /root/reference):
  scikit-learn gaussian_process/kernels.py (container 1.7.2):
    RBF.__call__        :1525-1566  cdist(X/l, Y/l, 'sqeuclidean'); exp(-0.5 D)
    Matern.__call__     :1680-1725  nu=2.5: t=sqrt(5) r; (1+t+t^2/3) exp(-t)
    DotProduct.__call__ :2162-2170  inner(X, Y) + sigma_0^2
"""
import math

import numpy as np
import scipy.linalg
from scipy.spatial.distance import cdist


# --------------------------------------------------------------------------
# kernel functions (reference: regressors.py:15-30 -> sklearn kernels)
# --------------------------------------------------------------------------
def _ls(length_scale, d):
    ls = np.asarray(length_scale, dtype=np.float64).reshape(-1)
    if ls.size not in (1, d):
        # sklearn _check_length_scale raises ValueError for a dimension mismatch
        raise ValueError(f"Anisotropic kernel must have the same number of dimensions as data ({ls.size}!={d})")
    return ls


def rbf_kernel(A, B, length_scale):
    """k(a,b)=exp(-0.5*||(a-b)/l||^2); sklearn RBF.__call__ (kernels.py:1556-1557)."""
    A = np.atleast_2d(A)
    B = np.atleast_2d(B)
    ls = _ls(length_scale, A.shape[1])
    D = cdist(A / ls, B / ls, metric="sqeuclidean")
    return np.exp(-0.5 * D)


def matern52_kernel(A, B, length_scale):
    """Matern nu=2.5; sklearn Matern.__call__ (kernels.py:1700-1711)."""
    A = np.atleast_2d(A)
    B = np.atleast_2d(B)
    ls = _ls(length_scale, A.shape[1])
    r = cdist(A / ls, B / ls, metric="euclidean")
    t = r * math.sqrt(5)
    return (1.0 + t + t ** 2 / 3.0) * np.exp(-t)


def linear_kernel(A, B, sigma_0):
    """DotProduct; sklearn kernels.py:2162-2170."""
    return np.inner(np.atleast_2d(A), np.atleast_2d(B)) + sigma_0 ** 2


class _K:
    """Tiny stand-in for the sklearn kernel object: exposes `.kernel(A, B)`."""

    def __init__(self, fn):
        self.kernel = fn


class ThreeDimensionalKernel(_K):
    """regressors.py:15-22 -- anisotropic RBF, length scales cycled (lx,ly,lz) over the state index."""

    def __init__(self, lx, ly, lz, n_states):
        l = [lx, ly, lz]
        self.length_scale = np.array([l[i % 3] for i in range(n_states)], dtype=np.float64)
        super().__init__(lambda A, B: rbf_kernel(A, B, self.length_scale))


class KernelWrapper(_K):
    """regressors.py:24-26 -- Matern(ls, nu=2.5)."""

    def __init__(self, ls):
        self.length_scale = np.asarray(ls, dtype=np.float64).reshape(-1)
        super().__init__(lambda A, B: matern52_kernel(A, B, self.length_scale))


class LinearKernelWrapper(_K):
    """regressors.py:28-30 -- DotProduct(sigma_0)."""

    def __init__(self, sigma):
        self.sigma_0 = float(sigma)
        super().__init__(lambda A, B: linear_kernel(A, B, self.sigma_0))


# --------------------------------------------------------------------------
# estimator (reference: regressors.py:32-55, 114-178)
# --------------------------------------------------------------------------
def _sqrtm_eigh(K):
    w, V = scipy.linalg.eigh(K)
    s = np.sqrt(w)
    return (V * s) @ V.T, (V / s) @ V.T


class KoopmanNystromOracle:
    """Restatement of KoopmanNystromRegressor (regressors.py:114-178) + predict (:48-55)."""

    def __init__(self, n_inputs, kernel=None, gamma=None, m=None, faithful=True):
        self.n_inputs = n_inputs
        self.kernel = kernel
        self.gamma = gamma
        self.m = m
        self.A = self.B = self.C = self.weights = None
        self.nystrom_centers_input = None
        self.nystrom_centers_output = None
        self.jitter = 1e-6
        self.faithful = faithful
        self.stages = {}

    # regressors.py:122-169
    def fit(self, X, Y):
        import time
        t0 = time.perf_counter()
        tm = self.timings = dict(kernel_n=0.0, gram_n=0.0, fixed=0.0)
        X = X.T
        Y = Y.T
        n_states = X.shape[0] - self.n_inputs
        gamma_n = self.gamma * X.shape[1]
        if self.nystrom_centers_output is None:  # :129-132 (global legacy RNG)
            idx = np.random.choice(np.arange(0, Y.shape[1]), size=self.m, replace=False)
            self.nystrom_centers_output = Y[:, idx]
        if self.nystrom_centers_input is None:  # :133-134
            self.nystrom_centers_input = self.nystrom_centers_output
        k = self.kernel.kernel
        Zo, Zi = self.nystrom_centers_output, self.nystrom_centers_input
        eye_m = np.eye(self.m)
        K_mm_out = k(Zo.T, Zo.T) + self.jitter * eye_m  # :139
        if self.faithful:
            S = scipy.linalg.sqrtm(K_mm_out).real  # :140
            Sinv = None
        else:
            S, Sinv = _sqrtm_eigh(K_mm_out)
        t1 = time.perf_counter()
        K_mn_out = k(Zo.T, Y.T)  # :141
        K_mn_in_x = k(Zi.T, X[:n_states, :].T)  # :142
        tm["kernel_n"] += time.perf_counter() - t1
        K_mm_in_x = k(Zi.T, Zi.T) + self.jitter * eye_m  # :143
        K_mm_in_x_out = k(Zi.T, Zo.T)  # :144 (no jitter)
        t1 = time.perf_counter()
        K_mn_in = np.vstack((K_mn_in_x, X[n_states:, :]))  # :147
        K_mm_in = scipy.linalg.block_diag(K_mm_in_x, np.eye(self.n_inputs))  # :148
        inner = K_mn_in @ K_mn_in.T + gamma_n * K_mm_in  # :151
        cross = K_mn_out @ K_mn_in.T
        tm["gram_n"] += time.perf_counter() - t1
        if self.faithful:
            right = scipy.linalg.block_diag(
                scipy.linalg.solve(S, K_mm_in_x_out.T, assume_a="her").T, np.eye(self.n_inputs))  # :152
            left = scipy.linalg.solve(S, cross, assume_a="her")  # :153
            sol = scipy.linalg.lstsq(inner, right)[0]  # :155
        else:
            right = scipy.linalg.block_diag((Sinv @ K_mm_in_x_out.T).T, np.eye(self.n_inputs))
            left = Sinv @ cross
            sol = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner), right)
        G = left @ sol  # :156
        self.A = G[:, : self.m]  # :158
        self.B = G[:, self.m:]  # :159
        t1 = time.perf_counter()
        inner_rec = gamma_n * K_mm_out + K_mn_out @ K_mn_out.T  # :162
        left_rec = Y @ K_mn_out.T  # :164
        tm["gram_n"] += time.perf_counter() - t1
        right_rec = scipy.linalg.sqrtm(K_mm_out).real if self.faithful else S  # :163
        if self.faithful:
            sol_rec = scipy.linalg.lstsq(inner_rec, right_rec)[0]  # :165
        else:
            sol_rec = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner_rec), right_rec)
        self.C = left_rec @ sol_rec  # :166
        self.weights = self.C @ G  # :167-169
        self._S, self._Sinv = S, Sinv
        tm["total"] = time.perf_counter() - t0
        tm["fixed"] = tm["total"] - tm["kernel_n"] - tm["gram_n"]  # everything that does not grow with n
        # intermediates kept for stage-level parity tests of the HIP path
        self.stages = dict(K_mm=K_mm_out, S=S, inner=inner, cross=cross, inner_rec=inner_rec,
                           left_rec=left_rec)

    # regressors.py:171-178
    def lift(self, X):
        k = self.kernel.kernel
        Zo = self.nystrom_centers_output
        Kmn = k(Zo.T, X.T)
        if self.faithful:
            Kmm = k(Zo.T, Zo.T) + self.jitter * np.eye(self.m)
            S = scipy.linalg.sqrtm(Kmm).real
            return scipy.linalg.solve(S, Kmn, assume_a="her")
        if getattr(self, "_Sinv", None) is None:
            Kmm = k(Zo.T, Zo.T) + self.jitter * np.eye(self.m)
            self._S, self._Sinv = _sqrtm_eigh(Kmm)
        return self._Sinv @ Kmn

    # regressors.py:48-55
    def predict(self, X_aug):
        n_states = X_aug.shape[1] - self.n_inputs
        X = X_aug.T
        phi = np.vstack((self.lift(X[:n_states, :]), X[n_states:, :]))
        return (self.weights @ phi).T


# --------------------------------------------------------------------------
# callers either side of the fit (reference: benchmark_lqr_*.py)
# --------------------------------------------------------------------------
def rollout(A, B, C, z0, controls):
    """Open-loop lifted rollout of validate_dyn_sys (benchmark_lqr_cloth.py:23-32).

    z0: (m,1) or (m,), controls: (p, T).  Returns (simulated (d,T), lifted (m,T)); column 0 is z0 / C z0,
    column t+1 = A z_t + B u_t  (the last control column is not used, as in the reference loop :29-32).
    """
    T = controls.shape[1]
    z = np.asarray(z0, dtype=np.float64).reshape(-1, 1)
    zs = [z]
    for i in range(T - 1):
        z = A @ z + B @ controls[:, i].reshape(-1, 1)
        zs.append(z)
    Z = np.hstack(zs)
    return C @ Z, Z


def validate_dyn_sys(reg, true_trajectory, test_controls, relative=False):
    """benchmark_lqr_cloth.py:18-36 (absolute RMSE, :34) / benchmark_lqr_classic.py:23-41 (relative-%, :39)."""
    z0 = reg.lift(true_trajectory[:, 0].reshape(-1, 1))
    T = true_trajectory.shape[1]
    ctrl = test_controls[:, :T]
    if ctrl.shape[1] == T - 1:  # classic / hjb drivers: one control per transition (the loop of :33 uses all of them)
        ctrl = np.hstack((ctrl, np.zeros((ctrl.shape[0], 1))))  # rollout() ignores the last column
    sim, _ = rollout(reg.A, reg.B, reg.C, z0, ctrl)
    if relative:
        return np.sqrt(np.sum(np.square(true_trajectory - sim))) / np.sqrt(np.sum(np.square(sim))) * 100
    return np.sqrt(np.mean(np.square(true_trajectory - sim)))


def dlqr(A, B, Q, R):
    """Stand-in for control.dlqr (benchmark_lqr_cloth.py:262): K = (B'PB+R)^-1 B'PA with P from the DARE."""
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(B.T @ P @ B + R, B.T @ P @ A)
    return K, P


def cloth_lqr_gain(A, B, C, c=0.0075):
    """benchmark_lqr_cloth.py:238-240,262-263: Q=c*C'C symmetrised, R=I, rows permuted for the simulator."""
    R = np.eye(B.shape[1])
    Q = c * C.T @ C
    Q = (Q + Q.T) / 2
    K, _ = dlqr(A, B, Q, R)
    return K, K[[0, 3, 1, 4, 2, 5], :]


def lqr_closed_loop_lifted(A, B, C, K, phi0, phi_ref, num_steps):
    """Lifted closed loop of lqr_control (benchmark_lqr_cloth.py:79-84): u=K(phi_ref-phi); phi<-A phi+B u.

    Returns (visited (d, num_steps), u_ops (p, num_steps)).
    """
    phi = phi0.reshape(-1, 1)
    xs, us = [], []
    for _ in range(num_steps):
        u = K @ (phi_ref.reshape(-1, 1) - phi)
        us.append(u)
        xs.append(C @ phi)
        phi = A @ phi + B @ u
    return np.hstack(xs), np.hstack(us)


def lqr_control_cloth(A, B, C, K, phi0, phi_ref, initial_state, num_steps, n_states=192):
    """benchmark_lqr_cloth.py:69-104 after the two lifts (:73-74): the lifted loop with the CUMULATIVE input sequence
    seeded from the control nodes (:76-81), the per-axis split (:85-101) and the simulator's row order (:102).
    Returns (x_s, y_s, z_s, final_us)."""
    initial_state = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    phi_new = np.asarray(phi0, dtype=np.float64).reshape(-1, 1)
    phi_reference = np.asarray(phi_ref, dtype=np.float64).reshape(-1, 1)
    visited_states = initial_state
    u_s = initial_state[[168, 169, 170, 189, 190, 191], :]
    for _ in range(num_steps):
        u_op = K @ (phi_reference - phi_new)
        u_s = np.hstack((u_s, u_s[:, -1].reshape(-1, 1) + u_op))
        visited_states = np.hstack((visited_states, C @ phi_new))
        phi_new = A @ phi_new + B @ u_op
    n_nodes = n_states // 3
    x_s = np.zeros((n_nodes, visited_states.shape[1]))
    y_s = np.zeros_like(x_s)
    z_s = np.zeros_like(x_s)
    for i in range(n_states):
        (x_s, y_s, z_s)[i % 3][i // 3, :] = visited_states[i, :]
    final_us = np.vstack((u_s[0, :], u_s[3, :], u_s[1, :], u_s[4, :], u_s[2, :], u_s[5, :]))
    return x_s, y_s, z_s, final_us


def lqr_control_plant(reg, K, plant_step, initial_state, reference, num_steps):
    """benchmark_lqr_hjb.py:73-97: plant in the loop, the state re-lifted every step.  Returns (x_s, u_s (p, steps))."""
    x = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    phi_ref = reg.lift(np.asarray(reference, dtype=np.float64).reshape(-1, 1))
    phi = reg.lift(x)
    xs, us = [], []
    for _ in range(num_steps):
        u = K @ (phi_ref - phi)
        us.append(u.reshape(-1, 1))
        xs.append(x[0, 0])
        x = np.asarray(plant_step(x, u)).reshape(-1, 1)
        phi = reg.lift(x)
    return np.array(xs), np.hstack(us)


def hjb_optimal_controls(plant_step, x0, num_steps):
    """The analytic optimum u* = x^3 - x sqrt(1 + x^4) rolled through the plant (benchmark_lqr_hjb.py:302-308)."""
    x = np.array([[float(x0)]])
    out = []
    for _ in range(num_steps):
        u = x ** 3 - x * np.sqrt(1 + x ** 4)
        out.append(float(u.squeeze()))
        x = plant_step(x, u)
    return np.array(out)


def cloth_reference_state(initial_state, alpha=np.pi / 4, vertical_shift=0.0, horizontal_shift=0.0):
    """benchmark_lqr_cloth.py:241-255."""
    x = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    offset = np.zeros_like(x)
    z_top = x[-1]
    for i in range(x.shape[0]):
        if i % 3 == 1:
            r = abs(z_top - x[i + 1])
            offset[i] = r * np.sin(alpha) + horizontal_shift
        if i % 3 == 2:
            r = abs(z_top - x[i])
            offset[i] = r - r * np.cos(alpha) + vertical_shift
    return x + offset


def truncated_solve(P, R, rcond=None):
    """What scipy.linalg.lstsq(P, R) (gelsd, regressors.py:155,165) computes in exact arithmetic: the minimum-norm
    solution with singular values <= rcond * s_max dropped (rcond = machine epsilon by default), evaluated through
    LAPACK's full SVD.  On a system with a clean spectral gap this is THE well-defined answer; gelsd's own internal SVD
    sometimes keeps one rounding-level singular value of an exactly singular matrix (its computed value lands just above
    eps * s_max) and then returns garbage along that direction -- tests/golden/f9_rank_deficient.npz records such a case.
    Returns (X, rank)."""
    rcond = np.finfo(np.float64).eps if rcond is None else rcond
    U, s, Vt = np.linalg.svd(np.asarray(P, dtype=np.float64))
    keep = s > rcond * s[0]
    return (Vt[keep].T / s[keep]) @ (U[:, keep].T @ R), int(keep.sum())


def kfold_slices(n, n_splits=5):
    """sklearn KFold(n_splits, shuffle=False): contiguous folds, the first n % k folds one longer."""
    sizes = np.full(n_splits, n // n_splits, dtype=int)
    sizes[: n % n_splits] += 1
    out, cur = [], 0
    for s in sizes:
        out.append((cur, cur + s))
        cur += s
    return out


def neg_rmse_score(Y_true, Y_pred):
    """sklearn scorer 'neg_root_mean_squared_error' (uniform average of per-output RMSE), negated."""
    return -float(np.mean(np.sqrt(np.mean((Y_true - Y_pred) ** 2, axis=0))))


def cv_fold_score(make_reg, X, Y, fold, centers_idx):
    """One (candidate, fold) unit of learn_hyperparams (benchmark_lqr_cloth.py:52-65): fit on the training
    rows with injected landmarks (indices into the training rows), score on the held-out rows."""
    lo, hi = fold
    tr = np.r_[0:lo, hi:X.shape[0]]
    Xtr, Ytr = X[tr], Y[tr]
    reg = make_reg()
    reg.nystrom_centers_output = Ytr.T[:, centers_idx]
    reg.fit(Xtr, Ytr)
    return neg_rmse_score(Y[lo:hi], reg.predict(X[lo:hi]))


# --------------------------------------------------------------------------
# plants used to regenerate the small-d configs (reference: dynamical_systems.py)
# --------------------------------------------------------------------------
def _rk4_ref(f, x, u, Ts):
    """dynamical_systems.py:29-43: note k4 is evaluated at x + k1*Ts (not k3), as the reference does."""
    k1 = f(x, u)
    k2 = f(x + k1 * Ts / 2, u)
    k3 = f(x + k2 * Ts / 2, u)
    k4 = f(x + k1 * Ts, u)
    return x + (Ts / 6) * (k1 + 2 * k2 + 2 * k3 + k4)


def duffing_step(x, u, Ts):
    """dynamical_systems.py:25-27,45-48."""
    x = x.reshape(2, -1)
    f = lambda x, u: -np.vstack((-x[1, :], 0.5 * x[1, :] + x[0, :] * (4 * x[0, :] ** 2 - 1) - 0.5 * u))
    return _rk4_ref(f, x, u, Ts)


def hjb_step(x, u, Ts):
    """dynamical_systems.py:92-93,111-112."""
    return _rk4_ref(lambda x, u: -x ** 3 + u, x, u, Ts)


# --------------------------------------------------------------------------
# synthetic headline workload
# --------------------------------------------------------------------------
def make_c4(n=100000, d=384, p=6, m=2000, seed=1234):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((n, d))
    U = rng.standard_normal((n, p))
    Wt = rng.standard_normal((d, d)) * 0.9 / math.sqrt(d)
    Bt = rng.standard_normal((p, d)) * 0.1
    Y = np.tanh(S @ Wt) + U @ Bt
    X = np.hstack([S, U])
    np.random.seed(0)
    idx = np.random.choice(np.arange(n), size=m, replace=False)
    return X, Y, idx
