"""Host-side mirror of the reference's estimator surface (regressors.py:32-55, 114-178): same constructor, same
public attributes, same fit / lift / predict shapes and error behaviour, so that the reference's drivers
(GridSearchCV, validate_dyn_sys, lqr_control, pickling) run unchanged -- with the arithmetic done by the HIP
kernels of libnyskoop.so through ctypes.  The DARE (control.dlqr in the reference) stays on the host.
"""
import ctypes as C
import time

import os

import numpy as np
from sklearn.base import BaseEstimator

from . import _lib
from .lqr import dlqr


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "stride")


def linear_rollout(A, B, C, z0, controls, return_lifted=False):
    """Open-loop recursion z_{t+1} = A z_t + B u_t, x_t = C z_t on the device for explicit operators (nk_linear_rollout).

    z0: (m,) / (m, 1) with controls (p, T) -> simulated (d, T) [and lifted (m, T)]; column 0 is C z0 and the last control
    column is not used, exactly like the loop of validate_dyn_sys (benchmark_lqr_cloth.py:23-32).
    z0: (batch, m) with controls (batch, T, p) -> (batch, T, d) [and (batch, T, m)]."""
    ctx = _lib.get_context()
    A = np.ascontiguousarray(A, dtype=np.float64)
    C_ = np.ascontiguousarray(C, dtype=np.float64)
    m, d = A.shape[0], C_.shape[0]
    B = np.asarray(B, dtype=np.float64)
    B = np.ascontiguousarray(B.reshape(m, B.size // m))
    p = B.shape[1]
    if A.shape != (m, m) or C_.shape != (d, m):
        raise ValueError(f"operator shapes {A.shape}, {B.shape}, {C_.shape} do not fit together")
    controls = np.asarray(controls, dtype=np.float64)
    single = controls.ndim == 2
    if single:
        z0b = np.ascontiguousarray(np.asarray(z0, dtype=np.float64).reshape(1, m))
        U = np.ascontiguousarray(controls.T).reshape(1, controls.shape[1], p)
    else:
        z0b = np.ascontiguousarray(np.asarray(z0, dtype=np.float64).reshape(-1, m))
        U = np.ascontiguousarray(controls)
    batch, T = U.shape[0], U.shape[1]
    if U.shape != (batch, T, p) or z0b.shape[0] != batch:
        raise ValueError(f"controls have shape {U.shape}, expected {(z0b.shape[0], T, p)}")
    out_x = np.empty((batch, T, d))
    out_z = np.empty((batch, T, m)) if return_lifted else None
    _lib.check(ctx.lib.nk_linear_rollout(ctx.handle, A.ctypes.data, B.ctypes.data if p else None, C_.ctypes.data, m, d, p,
                                         z0b.ctypes.data, U.ctypes.data if p else None, T, batch, out_x.ctypes.data,
                                         None if out_z is None else out_z.ctypes.data))
    if single:
        return (out_x[0].T, out_z[0].T) if return_lifted else out_x[0].T
    return (out_x, out_z) if return_lifted else out_x


class KoopmanRegressor(BaseEstimator):
    """regressors.py:32-55."""

    def __init__(self, n_inputs, gamma, m=None):
        self.gamma = gamma
        self.m = m
        self.A = None
        self.B = None
        self.C = None
        self.weights = None
        self.n_inputs = n_inputs

    def lift(self, X):
        raise NotImplementedError

    def fit(self, X, Y):
        raise NotImplementedError

    def rollout(self, x0, controls, return_lifted=False):
        """Open-loop forecast of validate_dyn_sys (benchmark_lqr_cloth.py:23-32; benchmark_lqr_hjb.py:23-44) for any
        estimator that provides `lift`, `A`, `B`, `C`: x0 (d,) with controls (p, T) -> simulated (d, T)."""
        z0 = self.lift(np.asarray(x0, dtype=np.float64).reshape(-1, 1))
        return linear_rollout(self.A, self.B, self.C, z0, controls, return_lifted)

    def predict(self, X_aug):
        """regressors.py:48-55 for a subclass that only provides `lift` and `weights`: the product W [phi; u] runs
        on the device through nk_gemm."""
        n_states = X_aug.shape[1] - self.n_inputs
        X = np.asarray(X_aug).T
        phi_X = np.ascontiguousarray(np.vstack((self.lift(X[:n_states, :]), X[n_states:, :])))
        W = np.ascontiguousarray(self.weights, dtype=np.float64)
        ctx = _lib.get_context()
        out = np.empty((W.shape[0], phi_X.shape[1]))
        _lib.check(ctx.lib.nk_gemm(ctx.handle, 0, 0, W.shape[0], phi_X.shape[1], W.shape[1], 1.0, W.ctypes.data,
                                   W.shape[1], phi_X.ctypes.data, phi_X.shape[1], 0.0, out.ctypes.data, out.shape[1]))
        return out.T


def _fetched(name):
    """Operator attribute of a fitted Nystrom regressor: `fit` queues the device->host copies and returns; the first
    access waits for them (nk_model_wait), so a sweep that fits the next candidate right away overlaps the copies of
    one fit with the kernels of the next.  Assigning an operator bumps a version counter that is part of the key of the
    cached device model, so predict / rollout never run on stale device copies; an IN-PLACE edit (`reg.A[0, 0] = 1`)
    cannot be seen -- follow it with `reg.A = reg.A` (or `reg.invalidate_device_model()`)."""
    key = "_" + name

    def get(self):
        self._wait_fetch()
        return self.__dict__.get(key)

    def set_(self, value):
        self._wait_fetch()
        self.__dict__[key] = value
        self.__dict__["_ops_version"] = self.__dict__.get("_ops_version", 0) + 1

    return property(get, set_)


class KoopmanNystromRegressor(KoopmanRegressor):
    """regressors.py:114-178, MI355X-native.

    Differences a caller can observe (all additive): `lift`/`predict` reuse K_mm^{-1/2} computed at fit time instead
    of re-running sqrtm per call (regressors.py:174-175); `fit` accepts `row_ranges` (training rows of a K-fold split
    without copying) and device-resident inputs; `rollout`, `score_neg_rmse`, `closed_loop` and `solve_lqr` expose
    the callers' inner loops (benchmark_lqr_cloth.py:18-36, 52-57, 69-104, 238-263) as single calls.
    """

    A = _fetched("A")
    B = _fetched("B")
    C = _fetched("C")
    weights = _fetched("weights")
    # Arithmetic of the O(n m d) kernel blocks and O(n m^2) Gram contractions of `fit`: "f64" (the only mode that meets the
    # 1e-6 operator bar) or "f32" (BASELINE.json's stress configuration; include/nyskoop.h, nk_set_compute_dtype).  An
    # attribute, not a constructor argument: the constructor mirrors the reference's (sklearn clone / get_params), and a
    # clone starts from the default.
    compute_dtype = "f64"

    def __init__(self, n_inputs, kernel=None, gamma=None, m=None):
        self._fetching = False
        super().__init__(n_inputs, gamma, m)
        self.kernel = kernel
        self.nystrom_centers_input = None
        self.nystrom_centers_output = None
        self.jitter = 1e-6
        self._model = None
        self._model_key = None
        self._stats = None
        self._ops_version = 0

    # ------------------------------------------------------------------------------------------------ plumbing
    def invalidate_device_model(self):
        """Forget the device copy of landmarks and operators (rebuilt from the host attributes at the next use)."""
        self._wait_fetch()
        self._drop_model()
    def _wait_fetch(self):
        if self.__dict__.get("_fetching"):
            self.__dict__["_fetching"] = False
            if self.__dict__.get("_fetch_lazy"):  # fit(..., fetch=False): copy the operators now, on first use
                self.__dict__["_fetch_lazy"] = False
                ctx = _lib.get_context()
                G, Cm, Wm = self.__dict__["_A"].base, self.__dict__["_C"], self.__dict__["_weights"]
                _lib.check(ctx.lib.nk_model_get_ops(ctx.handle, self._model, G.ctypes.data, G.shape[1], Cm.ctypes.data,
                                                    Cm.shape[1], Wm.ctypes.data, Wm.shape[1]))
            else:
                _lib.check(_lib.load_library().nk_model_wait(self._model))

    def __getstate__(self):
        self._wait_fetch()
        state = dict(self.__dict__)
        state["_fetch_lazy"] = False
        state["_model"] = None  # device handles never travel (benchmark_lqr_cloth.py:266-267 pickles regressors)
        state["_model_key"] = None
        return state

    def __setstate__(self, state):
        for name in ("A", "B", "C", "weights"):  # states written before the operators became properties
            if name in state:
                state["_" + name] = state.pop(name)
        self.__dict__.update(state)
        self.__dict__["_fetching"] = False
        self.__dict__.setdefault("_model", None)
        self.__dict__.setdefault("_model_key", None)
        self.__dict__.setdefault("_stats", None)
        self.__dict__.setdefault("_ops_version", 0)

    def __del__(self):
        try:
            self._drop_model()
        except Exception:
            pass

    def _drop_model(self):
        if getattr(self, "_model", None):
            self.__dict__["_fetching"] = False  # nk_model_destroy waits for a pending fetch itself
            _lib.load_library().nk_model_destroy(self._model)
            self._model = None
            self._model_key = None

    def _ensure_model(self):
        """Device model for lift/predict/rollout; rebuilt from host copies after un-pickling or when a caller
        replaced the landmarks / operators by hand."""
        if self.nystrom_centers_output is None:
            raise RuntimeError("regressor has no landmarks: call fit first")
        key = self._ops_key()
        if self._model is not None and self._model_key == key:
            return self._model
        self._wait_fetch()
        self._drop_model()
        ctx = _lib.get_context()
        Z = np.ascontiguousarray(np.asarray(self.nystrom_centers_output, dtype=np.float64).T)  # m x d
        m, d = Z.shape
        kd, keep = self.kernel.kernel.desc(d)
        p = int(self.n_inputs)

        def ptr(a, shape):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != shape:
                raise ValueError(f"operator has shape {a.shape}, expected {shape}")
            return a.ctypes.data, a

        pa, ka = ptr(self.A, (m, m))
        pb, kb = ptr(self.B, (m, p))
        pc, kc = ptr(self.C, (d, m))
        pw, kw = ptr(self.weights, (d, m + p))
        h = C.c_void_p()
        rc = ctx.lib.nk_model_create(ctx.handle, C.byref(kd), Z.ctypes.data, d, m, d, p, float(self.jitter),
                                     pa, pb, pc, pw, C.byref(h))
        if rc == -1:
            raise ValueError(ctx.lib.nk_last_error().decode())
        _lib.check(rc)
        self._model, self._model_key = h, key
        return h

    # ------------------------------------------------------------------------------------------------ fit
    def _prepare(self, n, d, Y=None):
        """Landmarks (regressors.py:129-134) and kernel descriptor for a fit on d-dimensional states."""
        if self.nystrom_centers_output is None:  # regressors.py:129-132: global legacy NumPy RNG, n = #samples
            if Y is None:
                raise RuntimeError("landmarks must be set before a fit from Gram blocks")
            idx = np.random.choice(np.arange(0, n), size=self.m, replace=False)
            if _is_device_tensor(Y):
                rows = Y[idx.tolist()]
                self.nystrom_centers_output = np.ascontiguousarray(rows.cpu().numpy().T)
            else:
                self.nystrom_centers_output = np.asarray(Y).T[:, idx]
        if self.nystrom_centers_input is None:  # regressors.py:133-134
            self.nystrom_centers_input = self.nystrom_centers_output
        Zo = np.ascontiguousarray(np.asarray(self.nystrom_centers_output, dtype=np.float64).T)
        if Zo.shape[1] != d:
            raise ValueError(f"landmarks have dimension {Zo.shape[1]}, data has {d}")
        same = self.nystrom_centers_input is self.nystrom_centers_output
        Zi = Zo if same else np.ascontiguousarray(np.asarray(self.nystrom_centers_input, dtype=np.float64).T)
        kd, keep = self.kernel.kernel.desc(d)
        return Zo, Zi, same, kd, keep

    @staticmethod
    def _ranges(row_ranges):
        if row_ranges is None:
            return None, 0, None
        flat = np.ascontiguousarray(np.asarray(row_ranges, dtype=np.int64).reshape(-1))
        return flat.ctypes.data_as(C.POINTER(C.c_int64)), flat.size // 2, flat

    @staticmethod
    def _raise(ctx, rc):
        if rc == -1:
            raise ValueError(ctx.lib.nk_last_error().decode())
        if rc == -3:
            raise np.linalg.LinAlgError(ctx.lib.nk_last_error().decode())
        _lib.check(rc)

    def _adopt(self, ctx, h, stats, m, d, p, t_host, fetch=True):
        """Take over a freshly fitted device model: queue the copies of the operators into page-locked arrays
        (fetch=False: leave them on the device until an attribute is read -- sweeps that only score never pay the copy)."""
        t_host2 = time.perf_counter()
        self._model = h
        self._stats = stats.as_dict()
        if fetch:
            G = _lib.pinned_empty((m, m + p))  # page-locked: the device->host copies run at the PCIe rate
            Cm = _lib.pinned_empty((d, m))
            Wm = _lib.pinned_empty((d, m + p))
        else:
            G, Cm, Wm = np.empty((m, m + p)), np.empty((d, m)), np.empty((d, m + p))
        t_host2b = time.perf_counter()
        self.__dict__.update(_A=G[:, :m], _B=G[:, m:], _C=Cm, _weights=Wm)  # A, B: views of G_ls (regressors.py:158-159)
        if fetch:
            _lib.check(ctx.lib.nk_model_get_ops_async(ctx.handle, h, G.ctypes.data, m + p, Cm.ctypes.data, m,
                                                      Wm.ctypes.data, m + p))
        self.__dict__["_fetch_lazy"] = not fetch
        self._fetching = True
        t_host3 = time.perf_counter()
        self._stats.update(host_ms_drop=(t_host[1] - t_host[0]) * 1e3, host_ms_call=(t_host2 - t_host[1]) * 1e3,
                           host_ms_fetch=(t_host3 - t_host2) * 1e3, host_ms_pinned=(t_host2b - t_host2) * 1e3)
        self._model_key = self._ops_key()

    def fit(self, X, Y, row_ranges=None, fetch=True):
        """regressors.py:122-169.  X: n x (d+p) rows [state | input], Y: n x d (NumPy arrays, or float64 device
        tensors already resident in HBM).  Returns None, like the reference."""
        ctx = _lib.get_context()
        Xm, Ym = _lib.Mat(X), _lib.Mat(Y)
        n, d = Ym.shape
        p = int(self.n_inputs)
        if Xm.shape != (n, d + p):
            raise ValueError(f"X has shape {Xm.shape}, expected {(n, d + p)}")
        Zo, Zi, same, kd, keep = self._prepare(n, d, Y)
        m = Zo.shape[0]
        rr, n_rr, keep_rr = self._ranges(row_ranges)
        stats = _lib.FitStats()
        h = C.c_void_p()
        t_host0 = time.perf_counter()
        self._drop_model()
        t_host1 = time.perf_counter()
        ctx.wait_for(X, Y)  # device tensors: whatever torch still has queued for them comes first
        ctx.set_compute_dtype(self.compute_dtype)
        try:
            rc = ctx.lib.nk_nystrom_fit(ctx.handle, C.byref(kd), Xm.ptr, Xm.ld, Ym.ptr, Ym.ld, n, d, p, rr, n_rr,
                                        None if same else Zi.ctypes.data, d, Zo.ctypes.data, d, m,
                                        float(self.gamma), float(self.jitter), C.byref(h), C.byref(stats))
        finally:
            if self.compute_dtype != "f64":
                ctx.set_compute_dtype("f64")
        self._raise(ctx, rc)
        self._adopt(ctx, h, stats, m, d, p, (t_host0, t_host1), fetch)

    # ------------------------------------------------------------------------- sample-sharded fit
    def gram_size(self, d):
        """Number of float64 entries of the packed Gram accumulator for d-dimensional states."""
        m = np.asarray(self.nystrom_centers_output).shape[1] if self.nystrom_centers_output is not None else int(self.m)
        cnt = C.c_int64()
        _lib.check(_lib.load_library().nk_gram_doubles(m, d, int(self.n_inputs), C.byref(cnt)))
        return int(cnt.value)

    def gram_partial(self, X, Y, row_ranges=None, out=None):
        """The four Gram blocks of regressors.py:151,153,162,164 (without the regularisers) over the given rows only,
        packed into one flat float64 buffer `out` (a NumPy array, or a 1-D device tensor which is then filled in
        place without leaving the GPU).  Landmarks must be set: every shard of a sharded fit uses the same ones."""
        ctx = _lib.get_context()
        Xm, Ym = _lib.Mat(X), _lib.Mat(Y)
        n, d = Ym.shape
        p = int(self.n_inputs)
        if Xm.shape != (n, d + p):
            raise ValueError(f"X has shape {Xm.shape}, expected {(n, d + p)}")
        if self.nystrom_centers_output is None:
            raise RuntimeError("landmarks must be set before gram_partial (all shards share them)")
        Zo, Zi, same, kd, keep = self._prepare(n, d)
        m = Zo.shape[0]
        cnt = self.gram_size(d)
        if out is None:
            out = np.empty(cnt)
        if _is_device_tensor(out):
            if out.dim() != 1 or out.numel() != cnt or not out.is_contiguous() or "float64" not in str(out.dtype):
                raise ValueError(f"out must be a contiguous float64 tensor with {cnt} entries")
            optr = out.data_ptr()
        else:
            if out.shape != (cnt,) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError(f"out must be a contiguous float64 array with {cnt} entries")
            optr = out.ctypes.data
        rr, n_rr, keep_rr = self._ranges(row_ranges)
        stats = _lib.FitStats()
        ctx.wait_for(X, Y, out)
        rc = ctx.lib.nk_nystrom_gram(ctx.handle, C.byref(kd), Xm.ptr, Xm.ld, Ym.ptr, Ym.ld, n, d, p, rr, n_rr,
                                     None if same else Zi.ctypes.data, d, Zo.ctypes.data, d, m, optr, C.byref(stats))
        self._raise(ctx, rc)
        self._gram_stats = stats.as_dict()
        return out

    def fit_from_gram(self, gram, n_total, d):
        """Finish the fit from accumulated Gram blocks (the sum of gram_partial over all shards); n_total = number of
        training rows over all shards (the `n` of regressors.py:127)."""
        ctx = _lib.get_context()
        p = int(self.n_inputs)
        Zo, Zi, same, kd, keep = self._prepare(n_total, d)
        m = Zo.shape[0]
        cnt = self.gram_size(d)
        if _is_device_tensor(gram):
            if gram.numel() != cnt or not gram.is_contiguous() or "float64" not in str(gram.dtype):
                raise ValueError(f"gram must be a contiguous float64 tensor with {cnt} entries")
            gptr = gram.data_ptr()
        else:
            gram = np.ascontiguousarray(gram, dtype=np.float64).reshape(-1)
            if gram.size != cnt:
                raise ValueError(f"gram must have {cnt} entries")
            gptr = gram.ctypes.data
        stats = _lib.FitStats()
        h = C.c_void_p()
        t_host0 = time.perf_counter()
        self._drop_model()
        t_host1 = time.perf_counter()
        ctx.wait_for(gram)  # e.g. the output of an all-reduce still running on torch's (RCCL's) stream
        rc = ctx.lib.nk_nystrom_solve(ctx.handle, C.byref(kd), None if same else Zi.ctypes.data, d, Zo.ctypes.data, d, m,
                                      d, p, gptr, int(n_total), float(self.gamma), float(self.jitter), C.byref(h),
                                      C.byref(stats))
        self._raise(ctx, rc)
        self._adopt(ctx, h, stats, m, d, p, (t_host0, t_host1))

    def _ops_key(self):
        # landmarks are a plain attribute (the reference's callers assign them): identity + shape; operators: the
        # version counter bumped by every assignment (ids alone could be reused by a new array after a free)
        z = self.nystrom_centers_output
        return (id(z), None if z is None else np.shape(z), self.__dict__.get("_ops_version", 0))

    # ------------------------------------------------------------------------------------------------ lift / predict
    def lift(self, X):
        """regressors.py:171-178: X is d x n_q (not augmented); returns phi, m x n_q."""
        ctx = _lib.get_context()
        h = self._ensure_model()
        Xq = _lib.Mat(X.t() if _is_device_tensor(X) else np.asarray(X, dtype=np.float64).T)
        nq = Xq.shape[0]
        m = np.asarray(self.nystrom_centers_output).shape[1]
        out = np.empty((nq, m))
        ctx.wait_for(X)
        _lib.check(ctx.lib.nk_lift(ctx.handle, h, Xq.ptr, Xq.ld, nq, out.ctypes.data, m))
        return out.T

    def predict(self, X_aug):
        """regressors.py:48-55: X_aug is n_q x (d+p); returns n_q x d."""
        ctx = _lib.get_context()
        h = self._ensure_model()
        Xm = _lib.Mat(X_aug)
        d = np.asarray(self.nystrom_centers_output).shape[0]
        if Xm.shape[1] != d + int(self.n_inputs):
            raise ValueError(f"X_aug has {Xm.shape[1]} columns, expected {d + int(self.n_inputs)}")
        out = np.empty((Xm.shape[0], d))
        ctx.wait_for(X_aug)
        _lib.check(ctx.lib.nk_predict(ctx.handle, h, Xm.ptr, Xm.ld, Xm.shape[0], out.ctypes.data, d))
        return out

    def score_neg_rmse(self, X_aug, Y):
        """sklearn's 'neg_root_mean_squared_error' of predict(X_aug) against Y, reduced on the device
        (benchmark_lqr_cloth.py:55)."""
        ctx = _lib.get_context()
        h = self._ensure_model()
        Xm, Ym = _lib.Mat(X_aug), _lib.Mat(Y)
        d = np.asarray(self.nystrom_centers_output).shape[0]
        if Xm.shape[1] != d + int(self.n_inputs):
            raise ValueError(f"X_aug has {Xm.shape[1]} columns, expected {d + int(self.n_inputs)}")
        if Ym.shape != (Xm.shape[0], d):
            raise ValueError(f"Y has shape {Ym.shape}, expected {(Xm.shape[0], d)}")
        if Xm.shape[0] == 0:
            raise ValueError("score_neg_rmse needs at least one row")
        s = C.c_double()
        ctx.wait_for(X_aug, Y)
        _lib.check(ctx.lib.nk_score_neg_rmse(ctx.handle, h, Xm.ptr, Xm.ld, Ym.ptr, Ym.ld, Xm.shape[0], C.byref(s)))
        return s.value

    # ------------------------------------------------------------------------------------------------ callers' loops
    def rollout(self, x0, controls, return_lifted=False):
        """Open-loop forecast of validate_dyn_sys (benchmark_lqr_cloth.py:23-32).

        x0: (d,) / (d,1) with controls (p, T)  -> simulated (d, T) [and lifted (m, T)];
        x0: (batch, d) with controls (batch, T, p) -> (batch, T, d) [and (batch, T, m)]."""
        ctx = _lib.get_context()
        h = self._ensure_model()
        d, m = np.asarray(self.nystrom_centers_output).shape
        p = int(self.n_inputs)
        controls = np.asarray(controls, dtype=np.float64)
        single = controls.ndim == 2
        if single:
            x0b = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(1, d))
            U = np.ascontiguousarray(controls.T).reshape(1, controls.shape[1], p)
        else:
            x0b = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1, d))
            U = np.ascontiguousarray(controls)
        batch, T = U.shape[0], U.shape[1]
        out_x = np.empty((batch, T, d))
        out_z = np.empty((batch, T, m)) if return_lifted else None
        _lib.check(ctx.lib.nk_rollout(ctx.handle, h, x0b.ctypes.data, d, U.ctypes.data, T, batch, out_x.ctypes.data,
                                      None if out_z is None else out_z.ctypes.data))
        if single:
            return (out_x[0].T, out_z[0].T) if return_lifted else out_x[0].T
        return (out_x, out_z) if return_lifted else out_x

    def closed_loop(self, K, phi0, phi_ref, num_steps):
        """Lifted closed loop of lqr_control (benchmark_lqr_cloth.py:79-84).

        phi0, phi_ref: (m,) / (m, 1)  -> (visited (d, steps), u_ops (p, steps));
        phi0, phi_ref: (batch, m)     -> (visited (batch, steps, d), u_ops (batch, steps, p)): independent loops that share
        the gain (several initial states / references in one call)."""
        ctx = _lib.get_context()
        h = self._ensure_model()
        d, m = np.asarray(self.nystrom_centers_output).shape
        p = int(self.n_inputs)
        K = np.ascontiguousarray(K, dtype=np.float64)
        if K.shape != (p, m):
            raise ValueError(f"gain has shape {K.shape}, expected {(p, m)}")
        phi0 = np.asarray(phi0, dtype=np.float64)
        phi_ref = np.asarray(phi_ref, dtype=np.float64)
        single = phi0.ndim < 2 or phi0.shape[1] == 1  # a vector or the (m, 1) column that `lift` returns
        if single:
            f0 = np.ascontiguousarray(phi0.reshape(1, m))
            fr = np.ascontiguousarray(phi_ref.reshape(1, m))
        else:
            f0 = np.ascontiguousarray(phi0.reshape(-1, m))
            fr = np.ascontiguousarray(np.broadcast_to(phi_ref.reshape(-1, m), f0.shape))
        batch = f0.shape[0]
        ox, ou = np.empty((batch, num_steps, d)), np.empty((batch, num_steps, p))
        _lib.check(ctx.lib.nk_closed_loop_batch(ctx.handle, h, K.ctypes.data, f0.ctypes.data, fr.ctypes.data,
                                                int(num_steps), batch, ox.ctypes.data, ou.ctypes.data))
        if single:
            return ox[0].T, ou[0].T
        return ox, ou

    def solve_lqr(self, Q=None, R=None, c=0.0075):
        """Host DARE, standing in for control.dlqr(A, B, Q, R) (benchmark_lqr_cloth.py:238-240,262): by default
        Q = c * C^T C symmetrised and R = I.  Returns the gain K (p x m)."""
        if Q is None:
            Q = c * self.C.T @ self.C
            Q = (Q + Q.T) / 2
        if R is None:
            R = np.eye(int(self.n_inputs))
        K, _ = dlqr(self.A, self.B, Q, R)
        return K

    @property
    def fit_stats_(self):
        return self._stats


class KoopmanKernelRegressor(KoopmanRegressor):
    """regressors.py:58-111: the exact (non-Nystrom) kernel estimator, lifted dimension N = #samples.  Used by the
    reference only as the accuracy comparator of benchmark_lqr_hjb.py:334-381 (N ~ 4000).  Composed on the host from
    the library's device building blocks (nk_kernel_matrix, nk_sqrtm_spd, nk_solve_spd, nk_gemm): every O(N^2 d) and
    O(N^3) step runs on the GPU; the reference's pinv(sqrtm(K)) is the inverse square root the Newton-Schulz
    iteration returns directly.
    """

    def __init__(self, n_inputs, kernel=None, gamma=None):
        super().__init__(n_inputs, gamma)
        self.kernel = kernel
        self.training_inputs = None
        self.training_outputs = None
        self.jitter = 1e-6

    @staticmethod
    def _gemm(ctx, A, B):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        out = np.empty((A.shape[0], B.shape[1]))
        _lib.check(ctx.lib.nk_gemm(ctx.handle, 0, 0, A.shape[0], B.shape[1], A.shape[1], 1.0, A.ctypes.data, A.shape[1],
                                   B.ctypes.data, B.shape[1], 0.0, out.ctypes.data, out.shape[1]))
        return out

    @staticmethod
    def _solve_spd(ctx, P, R):
        P = np.ascontiguousarray(P, dtype=np.float64)
        R = np.ascontiguousarray(R, dtype=np.float64)
        X = np.empty_like(R)
        rc = ctx.lib.nk_solve_spd(ctx.handle, P.ctypes.data, P.shape[1], P.shape[0], R.ctypes.data, R.shape[1],
                                  R.shape[1], X.ctypes.data, X.shape[1])
        if rc == -3:
            raise np.linalg.LinAlgError(ctx.lib.nk_last_error().decode())
        _lib.check(rc)
        return X

    # ---- device-resident composition: every N x N intermediate lives in HBM (torch tensors as the allocator), only the
    #      attributes the reference exposes are copied to the host, once, at the end.  At config 2's N = 1e4 an N x N matrix
    #      is 800 MB: the host-composed version below moves sixteen of them over PCIe.
    @staticmethod
    def _dgemm(ctx, A, B, out, tA=0, tB=0, beta=0.0):
        a, b, o = _lib.Mat(A), _lib.Mat(B), _lib.Mat(out)
        M, K = (a.shape[1], a.shape[0]) if tA else a.shape
        N = b.shape[0] if tB else b.shape[1]
        ctx.wait_for(A, B, out)
        _lib.check(ctx.lib.nk_gemm(ctx.handle, int(tA), int(tB), M, N, K, 1.0, a.ptr, a.ld, b.ptr, b.ld, float(beta), o.ptr, o.ld))
        return out

    @staticmethod
    def _dsolve(ctx, P, R, out):
        pm, rm, om = _lib.Mat(P), _lib.Mat(R), _lib.Mat(out)
        ctx.wait_for(P, R, out)
        rc = ctx.lib.nk_solve_spd(ctx.handle, pm.ptr, pm.ld, pm.shape[0], rm.ptr, rm.ld, rm.shape[1], om.ptr, om.ld)
        if rc == -3:
            raise np.linalg.LinAlgError(ctx.lib.nk_last_error().decode())
        _lib.check(rc)
        return out

    def _fit_device(self, ctx, torch, Xs, Us, Ys, N, gamma_n):
        dev = torch.device("cuda", ctx.device)
        f64 = torch.float64
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        new = lambda r, c: torch.empty((r, c), dtype=f64, device=dev)
        k = self.kernel.kernel
        p = Us.shape[1]
        Xs_d, Ys_d = up(Xs), up(Ys)
        Us_d = up(Us) if p > 0 else None
        K_ins = k(Xs_d, Xs_d, out=new(N, N))  # :82-84
        if p > 0:
            self._dgemm(ctx, Us_d, Us_d, K_ins, tB=1, beta=1.0)
        K_ins.diagonal().add_(gamma_n)
        Kout = k(Ys_d, Ys_d, out=new(N, N))  # :85
        Kout.diagonal().add_(self.jitter)
        S, Sinv = new(N, N), new(N, N)
        it, res = C.c_int32(), C.c_double()
        ctx.wait_for(Kout)
        _lib.check(ctx.lib.nk_sqrtm_spd(ctx.handle, Kout.data_ptr(), N, N, S.data_ptr(), Sinv.data_ptr(), C.byref(it),
                                        C.byref(res)))  # :87-88 (sqrtm, pinv)
        Kxy = k(Xs_d, Ys_d, out=new(N, N))  # :90
        right = new(N, N + p)  # :91-92: [(Sinv Kxy^T)^T | U]
        self._dgemm(ctx, Kxy, Sinv, right[:, :N], tB=1)
        if p > 0:
            right[:, N:] = Us_d
        del Kxy
        sol = self._dsolve(ctx, K_ins, right, new(N, N + p))
        del right, K_ins
        G_ls = self._dgemm(ctx, S, sol, new(N, N + p))  # :93
        del sol
        Phi = self._dgemm(ctx, Kout, Sinv, new(N, N), tA=1, tB=1)  # :98 Phi = (Sinv Kout)^T
        PPt = self._dgemm(ctx, Phi, Phi, new(N, N), tB=1)
        PPt.diagonal().add_(gamma_n)
        sol2 = self._dsolve(ctx, PPt, Phi, new(N, N))
        del PPt, Phi
        Yt_d = up(self.training_outputs)  # d x N
        Cd = self._dgemm(ctx, Yt_d, sol2, new(Yt_d.shape[0], N))  # :99
        del sol2
        Wd = self._dgemm(ctx, Cd, G_ls, new(Cd.shape[0], N + p))  # :100-102
        torch.cuda.synchronize(dev)
        G_host = G_ls.cpu().numpy()
        self.A = G_host[:, :N]
        self.B = G_host[:, N:]
        self.C = Cd.cpu().numpy()
        self.weights = Wd.cpu().numpy()
        self.Kout = Kout.cpu().numpy()
        self.Kout_sqrt_inv = Sinv.cpu().numpy()
        self._dev_cache = dict(Sinv=Sinv, Ys=Ys_d, device=ctx.device)  # lift() multiplies from HBM

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_dev_cache", None)  # device tensors never travel
        return state

    def fit(self, X, Y):
        ctx = _lib.get_context()
        X = np.asarray(X, dtype=np.float64).T  # (d+p) x N, as regressors.py:67-68
        Y = np.asarray(Y, dtype=np.float64).T
        n_states = X.shape[0] - self.n_inputs
        N = X.shape[1]
        gamma_n = self.gamma * N
        if self.training_inputs is None:
            self.training_inputs = X
        if self.training_outputs is None:
            self.training_outputs = Y
        k = self.kernel.kernel
        Xs = np.ascontiguousarray(self.training_inputs[:n_states, :].T)
        Us = np.ascontiguousarray(self.training_inputs[n_states:, :].T)  # N x p
        Ys = np.ascontiguousarray(self.training_outputs.T)
        self.__dict__.pop("_dev_cache", None)
        torch = _lib.torch_if_cuda()
        if torch is not None and os.environ.get("NYSKOOP_EXACT_HOST", "0") != "1":
            return self._fit_device(ctx, torch, Xs, Us, Ys, N, gamma_n)
        K_ins = k(Xs, Xs) + self._gemm(ctx, Us, Us.T) + gamma_n * np.eye(N)  # :82-84
        Kout = k(Ys, Ys) + self.jitter * np.eye(N)  # :85
        self.Kout = Kout
        S, Sinv = np.empty((N, N)), np.empty((N, N))
        it, res = C.c_int32(), C.c_double()
        _lib.check(ctx.lib.nk_sqrtm_spd(ctx.handle, Kout.ctypes.data, N, N, S.ctypes.data, Sinv.ctypes.data,
                                        C.byref(it), C.byref(res)))  # :87-88 (sqrtm, pinv)
        self.Kout_sqrt_inv = Sinv
        Kins_x_outs = k(Xs, Ys)  # :90
        right_state = self._gemm(ctx, Sinv, Kins_x_outs.T).T  # :91
        right = np.hstack((right_state, Us))  # :92
        G_ls = self._gemm(ctx, S, self._solve_spd(ctx, K_ins, right))  # :93
        self.A = G_ls[:, :N]
        self.B = G_ls[:, N:]
        # :98-99 -- Phi = (S^-1 Kout)^T; C = Y (Phi Phi^T + gamma_n I)^-1 Phi
        Phi = self._gemm(ctx, Sinv, Kout).T
        PPt = self._gemm(ctx, Phi, Phi.T) + gamma_n * np.eye(N)
        self.C = self._gemm(ctx, self.training_outputs, self._solve_spd(ctx, PPt, Phi))
        self.weights = self._gemm(ctx, self.C, G_ls)  # :100-102

    def lift(self, X):
        """regressors.py:104-111."""
        ctx = _lib.get_context()
        Xq = np.ascontiguousarray(np.asarray(X, dtype=np.float64).T)
        cache = self.__dict__.get("_dev_cache")
        if cache is not None and cache["device"] == ctx.device:
            import torch
            Sinv, Ys_d = cache["Sinv"], cache["Ys"]
            Xq_d = torch.from_numpy(Xq).to(Ys_d.device)
            Kt = self.kernel.kernel(Ys_d, Xq_d, out=torch.empty((Ys_d.shape[0], Xq.shape[0]), dtype=torch.float64, device=Ys_d.device))
            out = self._dgemm(ctx, Sinv, Kt, torch.empty_like(Kt))
            return out.cpu().numpy()
        Kout_test = self.kernel.kernel(np.ascontiguousarray(self.training_outputs.T), Xq)
        return self._gemm(ctx, self.Kout_sqrt_inv, Kout_test)
