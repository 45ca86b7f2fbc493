"""ctypes binding of libnyskoop.so (include/nyskoop.h).  Fails loudly: no library or no GPU => exception."""
import atexit
import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NK_KERNEL_RBF, NK_KERNEL_MATERN52, NK_KERNEL_LINEAR = 0, 1, 2
NK_OK = 0
_ERR_NAMES = {-1: "NK_ERR_BAD_ARG", -2: "NK_ERR_HIP", -3: "NK_ERR_NOT_SPD", -4: "NK_ERR_OOM",
              -5: "NK_ERR_NO_CONVERGENCE", -6: "NK_ERR_NO_DEVICE"}


class NyskoopError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class KernelDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("d", C.c_int32), ("n_lengthscale", C.c_int32), ("reserved", C.c_int32),
                ("lengthscale", C.POINTER(C.c_double)), ("sigma0", C.c_double)]


class CvUnit(C.Structure):
    _fields_ = [("kernel", C.POINTER(KernelDesc)), ("gamma", C.c_double), ("jitter", C.c_double), ("m", C.c_int32),
                ("reserved", C.c_int32), ("test_begin", C.c_int64), ("test_end", C.c_int64),
                ("landmark_rows", C.POINTER(C.c_int64))]


class FitStats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_upload", C.c_double), ("ms_kmat", C.c_double),
                ("ms_gram", C.c_double), ("ms_sqrt", C.c_double), ("ms_solve", C.c_double),
                ("ms_gram_kernel_avg", C.c_double), ("gram_kernel_launches", C.c_int32),
                ("sqrt_iters", C.c_int32), ("sqrt_residual", C.c_double), ("gram_flops", C.c_double),
                ("kmat_pairs", C.c_double), ("rank_inner", C.c_int32), ("rank_inner_rec", C.c_int32),
                ("pivot_ratio_inner", C.c_double), ("pivot_ratio_inner_rec", C.c_double), ("refined", C.c_int32),
                ("reserved_", C.c_int32), ("refine_ratio_inner", C.c_double), ("refine_ratio_inner_rec", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_D = C.c_double

# name -> (restype, argtypes); every symbol declared in include/nyskoop.h
SIGNATURES = {
    "nk_version": (C.c_int, []),
    "nk_last_error": (C.c_char_p, []),
    "nk_device_count": (C.c_int, []),
    "nk_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "nk_destroy": (C.c_int, [_P]),
    "nk_synchronize": (C.c_int, [_P]),
    "nk_stream": (_P, [_P]),
    "nk_set_kmat_mode": (C.c_int, [_P, C.c_int]),
    "nk_set_strict_spd": (C.c_int, [_P, C.c_int]),
    "nk_set_refine": (C.c_int, [_P, C.c_double, C.c_int32]),
    "nk_wait_stream": (C.c_int, [_P, _P]),
    "nk_shutdown": (C.c_int, []),
    "nk_group_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "nk_group_enter": (C.c_int, [_P]),
    "nk_group_leave": (C.c_int, [_P]),
    "nk_set_compute_dtype": (C.c_int, [_P, C.c_int]),
    "nk_group_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "nk_runtime_counters": (C.c_int, [C.POINTER(C.c_uint64), C.c_int32]),
    "nk_cv_grid": (C.c_int, [C.POINTER(_P), _I32, _P, _I64, _P, _I64, _I64, _I32, _I32, C.POINTER(CvUnit), _I32,
                             C.POINTER(_D), C.POINTER(_I32)]),
    "nk_host_alloc": (_P, [C.c_uint64]),
    "nk_host_free": (None, [_P]),
    "nk_kernel_matrix": (C.c_int, [_P, C.POINTER(KernelDesc), _P, _I64, _I64, _P, _I64, _I64, _P, _I64]),
    "nk_nystrom_fit": (C.c_int, [_P, C.POINTER(KernelDesc), _P, _I64, _P, _I64, _I64, _I32, _I32,
                                 C.POINTER(_I64), _I32, _P, _I64, _P, _I64, _I32, _D, _D,
                                 C.POINTER(_P), C.POINTER(FitStats)]),
    "nk_gram_doubles": (C.c_int, [_I32, _I32, _I32, C.POINTER(_I64)]),
    "nk_nystrom_gram": (C.c_int, [_P, C.POINTER(KernelDesc), _P, _I64, _P, _I64, _I64, _I32, _I32,
                                  C.POINTER(_I64), _I32, _P, _I64, _P, _I64, _I32, _P, C.POINTER(FitStats)]),
    "nk_nystrom_solve": (C.c_int, [_P, C.POINTER(KernelDesc), _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _D, _D,
                                   C.POINTER(_P), C.POINTER(FitStats)]),
    "nk_model_create": (C.c_int, [_P, C.POINTER(KernelDesc), _P, _I64, _I32, _I32, _I32, _D, _P, _P, _P, _P,
                                  C.POINTER(_P)]),
    "nk_model_destroy": (C.c_int, [_P]),
    "nk_model_get": (C.c_int, [_P, _P, C.c_char, _P, _I64]),
    "nk_model_dims": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "nk_model_get_ops": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _P, _I64]),
    "nk_model_get_ops_async": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _P, _I64]),
    "nk_model_wait": (C.c_int, [_P]),
    "nk_lift": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I64]),
    "nk_predict": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I64]),
    "nk_score_neg_rmse": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, C.POINTER(_D)]),
    "nk_rollout": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, _P, _P]),
    "nk_closed_loop": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P]),
    "nk_closed_loop_batch": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _P, _P]),
    "nk_linear_rollout": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _P]),
    "nk_gemm_f32": (C.c_int, [_P, _I64, _I64, _I64, _P, _I64, _P, _I64, _P, _I64]),
    "nk_gemm": (C.c_int, [_P, C.c_int, C.c_int, _I64, _I64, _I64, _D, _P, _I64, _P, _I64, _D, _P, _I64]),
    "nk_sqrtm_spd": (C.c_int, [_P, _P, _I64, _I32, _P, _P, C.POINTER(_I32), C.POINTER(_D)]),
    "nk_solve_spd": (C.c_int, [_P, _P, _I64, _I32, _P, _I64, _I32, _P, _I64]),
    "nk_bench_gram": (C.c_int, [_P, _I64, _I32, _I32, _I32, _I32, C.POINTER(_D), C.POINTER(_D)]),
}

_lib = None
_lib_lock = threading.Lock()


def library_path():
    return os.environ.get("NYSKOOP_LIB", os.path.join(_HERE, "libnyskoop.so"))


def load_library():
    """dlopen libnyskoop.so and declare every prototype.  Raises if the library has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            path = library_path()
            if not os.path.exists(path):
                raise NyskoopError(-6, f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
                                       f"g.build()'` or `make -C nys_koop_lqr_amd/csrc` (there is no CPU fallback)")
            # torch wheels bundle their own HIP runtime (same SONAME as /opt/rocm's): if this library pulled in the system
            # runtime first, a later `import torch` in the same process would find no GPUs.  When torch is installed,
            # let it load its runtime first (plumbing only: nothing here calls into torch).
            if os.environ.get("NYSKOOP_PRELOAD_TORCH", "1") == "1" and "torch" not in sys.modules:
                try:
                    import torch  # noqa: F401
                except Exception:
                    pass
            lib = C.CDLL(path)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
            # Deterministic teardown: release every stream, event, workspace, model buffer and page-locked block while
            # the HIP runtime (and a profiler's tool library, if one is attached) is still fully alive.  atexit hooks of
            # the interpreter run before Py_Finalize and before the C runtime's exit handlers / static destructors;
            # afterwards the finalisers of Context / regressor / pinned-array objects find their handles already
            # released (nk_destroy, nk_model_destroy and nk_host_free are no-ops for handles the library no longer knows).
            atexit.register(shutdown)
    return _lib


def shutdown():
    """nk_shutdown(): wait for pending work and release everything the library holds.  Idempotent; contexts, device
    models and pinned result arrays that are still referenced become invalid (operators already fetched stay valid only
    if they were copied: `np.array(reg.A)`)."""
    lib = _lib
    if lib is None:
        return
    with _pools_lock:
        pools = list(_pools.values())
        _pools.clear()
        lpools = list(_lockstep_pools.values())
        _lockstep_pools.clear()
    for pool in pools:  # worker threads hold contexts in thread-local storage; stop them before their streams go away
        pool.shutdown(wait=True)
    for pool in lpools:
        pool.close()
    _PinnedBlock.drain()
    lib.nk_shutdown()


def check(rc):
    if rc != NK_OK:
        raise NyskoopError(rc, load_library().nk_last_error().decode("utf-8", "replace"))


def runtime_counters():
    """Process-wide counts of the library's silent slow paths (nk_runtime_counters): single-launch recursions and Jacobi
    sweeps that gave up waiting for non-resident workgroups, fits that took the rank-truncating branch of the
    reference's lstsq (regressors.py:155,165), fits that repeated the matrix square root."""
    v = (C.c_uint64 * 5)()
    check(load_library().nk_runtime_counters(v, 5))
    return dict(chain_giveups=int(v[0]), jacobi_giveups=int(v[1]), rank_truncated_fits=int(v[2]), sqrt_retries=int(v[3]),
                refined_fits=int(v[4]))


def torch_if_cuda():
    """torch if it is importable and sees a HIP device (it is the allocator of device-resident intermediates where a
    caller-side composition wants them, e.g. KoopmanKernelRegressor.fit), else None."""
    try:
        import torch
    except Exception:  # noqa: BLE001 -- optional dependency
        return None
    try:
        return torch if torch.cuda.is_available() else None
    except Exception:  # noqa: BLE001
        return None


class Mat:
    """A row-major float64 matrix view handed to the C-ABI: host (numpy) or device (anything with
    data_ptr()/stride()/shape such as a torch.cuda tensor).  Keeps the owner alive for the call."""

    def __init__(self, obj, rows=None, cols=None):
        if hasattr(obj, "data_ptr") and hasattr(obj, "stride"):  # device tensor (duck-typed torch)
            if str(getattr(obj, "dtype", "")).split(".")[-1] not in ("float64", "double"):
                raise TypeError("device tensors must be float64")
            if obj.dim() != 2 or (obj.shape[1] > 1 and obj.stride(1) != 1):  # (a size-1 dimension may carry any stride)
                raise ValueError("device tensors must be 2-D with unit inner stride")
            self.owner, self.ptr, self.ld = obj, obj.data_ptr(), int(obj.stride(0))
            self.shape = (int(obj.shape[0]), int(obj.shape[1]))
        else:
            a = np.asarray(obj, dtype=np.float64)
            if a.ndim == 1:
                a = a.reshape(1, -1)
            if a.ndim != 2:
                raise ValueError("expected a 2-D array")
            if a.shape[1] > 0 and a.shape[0] > 0 and (a.strides[1] != 8 or a.strides[0] % 8 or a.strides[0] < 8 * a.shape[1]):
                a = np.ascontiguousarray(a)
            self.owner, self.ptr = a, a.ctypes.data
            self.ld = a.strides[0] // 8 if a.shape[0] > 1 else max(a.shape[1], 1)
            self.shape = a.shape
        if self.shape[0] <= 1:
            self.ld = max(self.ld, self.shape[1], 1)
        if rows is not None and self.shape[0] != rows or cols is not None and self.shape[1] != cols:
            raise ValueError(f"expected a {rows} x {cols} matrix, got {self.shape}")


class Context:
    """One nk_ctx (HIP stream + HBM workspace) on one device.  Not thread-safe; one per process and device."""

    def __init__(self, device=0, handle=None):
        self.lib = load_library()
        if handle is None:
            handle = _P()
            check(self.lib.nk_create(int(device), C.byref(handle)))
        self.handle = handle
        self.device = int(device)

    def enter(self):
        """Lock-step group member: start of a unit of work (nk_group_enter)."""
        check(self.lib.nk_group_enter(self.handle))

    def leave(self):
        check(self.lib.nk_group_leave(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.nk_synchronize(self.handle))

    @property
    def stream(self):
        return self.lib.nk_stream(self.handle)

    def set_strict_spd(self, strict):
        """True: a numerically rank-deficient regularised system raises LinAlgError (NK_ERR_NOT_SPD) instead of being
        solved like scipy.linalg.lstsq does (minimum-norm solution, singular values <= eps * sigma_max dropped).
        2 / "lstsq": every regularised system goes through the SVD with gelsd's cut-off, ill-conditioned or not."""
        check(self.lib.nk_set_strict_spd(self.handle, 2 if strict in (2, "lstsq") else (1 if strict else 0)))

    def set_refine(self, pivot_ratio=1e-9, steps=2):
        """Refine the regularised solves of fits whose smallest / largest Cholesky pivot is below `pivot_ratio` with
        doubled-precision residuals (nk_set_refine); pivot_ratio = 0 switches it off (the default)."""
        check(self.lib.nk_set_refine(self.handle, float(pivot_ratio), int(steps)))

    def wait_for(self, *objs):
        """Order the context's streams after the work queued on torch's current stream whenever one of `objs` is a
        device tensor: the library's streams are non-blocking, so nothing else makes them wait for, say, a pending
        all-reduce that produces the tensor (nk_wait_stream; the host does not block)."""
        torch = sys.modules.get("torch")
        if torch is None:
            return
        for o in objs:
            if hasattr(o, "data_ptr") and hasattr(o, "stride") and getattr(o, "is_cuda", False):
                stream = torch.cuda.current_stream(o.device)
                check(self.lib.nk_wait_stream(self.handle, C.c_void_p(stream.cuda_stream)))
                return

    def set_compute_dtype(self, dtype):
        """'f64' (default) or 'f32': arithmetic of the kernel blocks and Gram contractions of fits on this context
        (nk_set_compute_dtype: the stress configuration's fp32 engine; everything m x m stays fp64)."""
        code = {"f64": 0, "f32": 1}.get(str(dtype))
        if code is None:
            raise ValueError("dtype must be 'f64' or 'f32'")
        check(self.lib.nk_set_compute_dtype(self.handle, code))

    def set_kmat_mode(self, mode):
        """0 = automatic (Gram form on the MFMA engine for d >= 32), 1 = always direct differences."""
        check(self.lib.nk_set_kmat_mode(self.handle, int(mode)))


class _PinnedBlock:
    """A page-locked host block that returns to a small pool when its last array view dies."""
    _pool = {}
    _lock = threading.Lock()
    _closed = False

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = None
        with _PinnedBlock._lock:
            free = _PinnedBlock._pool.get(self.nbytes)
            if free:
                self.ptr = free.pop()
        if self.ptr is None:
            self.ptr = load_library().nk_host_alloc(self.nbytes)
            if not self.ptr:
                raise MemoryError(f"nk_host_alloc({self.nbytes}) failed")

    def __del__(self):
        try:
            with _PinnedBlock._lock:
                free = _PinnedBlock._pool.setdefault(self.nbytes, [])
                if not _PinnedBlock._closed and len(free) < 8:
                    free.append(self.ptr)
                    return
            load_library().nk_host_free(self.ptr)  # a no-op after nk_shutdown
        except Exception:
            pass

    @classmethod
    def drain(cls):
        """Free the pooled blocks and stop pooling (blocks still referenced by arrays are released by nk_shutdown)."""
        with cls._lock:
            ptrs = [p for free in cls._pool.values() for p in free]
            cls._pool.clear()
            cls._closed = True
        lib = load_library()
        for p in ptrs:
            lib.nk_host_free(p)


def pinned_empty(shape):
    """float64 C-contiguous ndarray backed by page-locked memory (results of fit land here)."""
    n = int(np.prod(shape))
    if n == 0:
        return np.empty(shape)
    block = _PinnedBlock(n * 8)
    buf = (C.c_double * n).from_address(block.ptr)
    buf._nk_owner = block  # the ctypes array is the ndarray's base and keeps the block alive
    return np.frombuffer(buf, dtype=np.float64).reshape(shape)


_tls = threading.local()


def default_device():
    return int(os.environ.get("NYSKOOP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def get_context(device=None):
    """The calling thread's context on `device` (a context is not thread-safe: one per thread and device; independent
    fits issued from several threads overlap on the GPU).  Contexts live in thread-local storage, so a worker thread's
    streams and workspace are released when the thread ends."""
    device = default_device() if device is None else int(device)
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    key = (os.getpid(), device)
    ctx = ctxs.get(key)
    if ctx is None:
        ctx = ctxs[key] = Context(device)
    return ctx


class LockstepPool:
    """`size` host threads, thread i permanently bound to member i of a lock-step group (nk_group_create): the units of a
    round run through the ordinary API, one per thread, and the library merges their kernel launches (see
    include/nyskoop.h).  run_round(fn, items) runs fn(item) for up to `size` items, one per member, and returns the results
    in order; an exception in a unit is re-raised after the round."""

    def __init__(self, size, device=None):
        import queue
        self.size = int(size)
        self.device = default_device() if device is None else int(device)
        lib = load_library()
        handles = (_P * self.size)()
        check(lib.nk_group_create(self.device, self.size, handles))
        self.members = [Context(self.device, _P(handles[i])) for i in range(self.size)]
        self._queues = [queue.Queue() for _ in range(self.size)]
        self._done = queue.Queue()
        self._threads = [threading.Thread(target=self._worker, args=(i,), daemon=True, name=f"nyskoop-lockstep-{i}")
                         for i in range(self.size)]
        for t in self._threads:
            t.start()

    def _worker(self, i):
        member = self.members[i]
        _tls.ctxs = {(os.getpid(), self.device): member}
        while True:
            task = self._queues[i].get()
            if task is None:
                return
            fn, item, k = task
            try:
                out = (k, fn(item), None)
            except BaseException as e:  # noqa: BLE001 -- reported to the caller of run_round
                out = (k, None, e)
            finally:
                try:
                    member.leave()
                except Exception:
                    pass
            self._done.put(out)

    def run_round(self, fn, items):
        items = list(items)
        if len(items) > self.size:
            raise ValueError("more items than members")
        for k in range(len(items)):  # all members of the round are inside their unit before any of them starts
            self.members[k].enter()
        for k, item in enumerate(items):
            self._queues[k].put((fn, item, k))
        results, err = [None] * len(items), None
        for _ in items:
            k, out, e = self._done.get()
            results[k] = out
            err = err or e
        if err is not None:
            raise err
        return results

    def map(self, fn, items):
        items = list(items)
        out = []
        for r in range(0, len(items), self.size):
            out.extend(self.run_round(fn, items[r:r + self.size]))
        return out

    def cv_grid(self, X, Y, n_inputs, units):
        """nk_cv_grid: `units` = list of (DeviceKernel, gamma, jitter, m, (test_begin, test_end), landmark_rows) run in
        lock step, len(members) at a time, entirely inside the library (no Python per unit).  Returns (scores, status)."""
        Xm, Ym = Mat(X), Mat(Y)
        n, d = Ym.shape
        p = int(n_inputs)
        if Xm.shape != (n, d + p):
            raise ValueError(f"X has shape {Xm.shape}, expected {(n, d + p)}")
        arr = (CvUnit * len(units))()
        keep = []
        descs = {}
        for i, (kern, gamma, jitter, m, (lo, hi), rows) in enumerate(units):
            if id(kern) not in descs:
                descs[id(kern)] = kern.desc(d)
            kd, ls = descs[id(kern)]
            rows = np.ascontiguousarray(rows, dtype=np.int64)
            if rows.shape != (int(m),):
                raise ValueError("landmark_rows must hold m row indices")
            keep.append(rows)
            arr[i].kernel = C.pointer(kd)
            arr[i].gamma, arr[i].jitter, arr[i].m = float(gamma), float(jitter), int(m)
            arr[i].test_begin, arr[i].test_end = int(lo), int(hi)
            arr[i].landmark_rows = rows.ctypes.data_as(C.POINTER(C.c_int64))
        handles = (_P * self.size)(*[m_.handle for m_ in self.members])
        scores = np.full(len(units), np.nan)
        status = np.zeros(len(units), dtype=np.int32)
        lib = self.members[0].lib
        rc = lib.nk_cv_grid(handles, self.size, Xm.ptr, Xm.ld, Ym.ptr, Ym.ld, n, d, p, arr, len(units),
                            scores.ctypes.data_as(C.POINTER(_D)), status.ctypes.data_as(C.POINTER(_I32)))
        if rc == -1:
            raise ValueError(lib.nk_last_error().decode())
        check(rc)
        return scores, status

    def stats(self):
        v = (C.c_uint64 * 4)()
        check(self.members[0].lib.nk_group_stats(self.members[0].handle, v))
        return dict(flushes=int(v[0]), merged_launches=int(v[1]), single_launches=int(v[2]), member_launches_merged=int(v[3]))

    def close(self):
        for q in self._queues:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)
        for m in self.members:
            m.close()


_lockstep_pools = {}


def lockstep_pool(size, device=None, index=0):
    """A persistent LockstepPool of `size` members on `device` (created on first use); `index` distinguishes several
    pools of the same size (independent groups whose work overlaps on the GPU)."""
    device = default_device() if device is None else int(device)
    with _pools_lock:
        key = (os.getpid(), device, int(size), int(index))
        pool = _lockstep_pools.get(key)
        if pool is None:
            pool = _lockstep_pools[key] = LockstepPool(size, device)
        return pool


_pools = {}
_pools_lock = threading.Lock()


def worker_pool(workers):
    """A persistent pool of `workers` host threads (their contexts, streams and workspaces are reused across calls)."""
    from concurrent.futures import ThreadPoolExecutor
    with _pools_lock:
        key = (os.getpid(), int(workers))
        pool = _pools.get(key)
        if pool is None:
            pool = _pools[key] = ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="nyskoop")
        return pool
