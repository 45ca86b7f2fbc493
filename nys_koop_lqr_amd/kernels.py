"""Kernel objects mirroring regressors.py:15-30 of the reference: each wrapper exposes `.kernel`, a callable
`(A (n_a x d), B (n_b x d)) -> (n_a x n_b)` -- here evaluated by the HIP kernel-matrix kernel through
nk_kernel_matrix instead of scikit-learn's RBF / Matern / DotProduct objects."""
import ctypes as C

import numpy as np

from . import _lib


class DeviceKernel:
    """Stands where the reference holds an sklearn kernel object (regressors.py:22,26,30)."""

    def __init__(self, ktype, length_scale=None, sigma_0=0.0):
        self.ktype = int(ktype)
        self.length_scale = None if length_scale is None else np.atleast_1d(
            np.squeeze(np.asarray(length_scale, dtype=np.float64))).copy()
        self.sigma_0 = float(sigma_0)

    def desc(self, d):
        """(KernelDesc, keep-alive) for data of dimension d.  Dimension mismatches are reported by the library
        with sklearn's message (_check_length_scale) and surface as ValueError."""
        kd = _lib.KernelDesc()
        kd.type, kd.d, kd.sigma0 = self.ktype, int(d), self.sigma_0
        ls = self.length_scale if self.length_scale is not None else np.ones(1)
        ls = np.ascontiguousarray(ls, dtype=np.float64)
        kd.n_lengthscale = int(ls.size)
        kd.lengthscale = ls.ctypes.data_as(C.POINTER(C.c_double))
        return kd, ls

    def __call__(self, X, Y=None, device=None, out=None):
        """K(X, Y) as a NumPy array; with `out` (a float64 device tensor of shape (len(X), len(Y))) the result stays in
        HBM and `out` is returned (X, Y may be device tensors as well)."""
        ctx = _lib.get_context(device)
        A = _lib.Mat(X)
        B = A if Y is None else _lib.Mat(Y)
        if A.shape[1] != B.shape[1]:
            raise ValueError(f"XA and XB must have the same number of columns ({A.shape[1]} != {B.shape[1]})")
        kd, keep = self.desc(A.shape[1])
        if out is None:
            res = np.empty((A.shape[0], B.shape[0]), dtype=np.float64)
            optr, old = res.ctypes.data, max(res.shape[1], 1)
        else:
            om = _lib.Mat(out, rows=A.shape[0], cols=B.shape[0])
            res, optr, old = out, om.ptr, om.ld
            ctx.wait_for(X, Y, out)
        rc = ctx.lib.nk_kernel_matrix(ctx.handle, C.byref(kd), A.ptr, A.ld, A.shape[0], B.ptr, B.ld, B.shape[0], optr, old)
        if rc == -1:
            raise ValueError(ctx.lib.nk_last_error().decode())
        _lib.check(rc)
        return res

    def __repr__(self):
        name = {0: "RBF", 1: "Matern52", 2: "DotProduct"}[self.ktype]
        return f"{name}(length_scale={self.length_scale}, sigma_0={self.sigma_0})"


class ThreeDimensionalKernel:
    """regressors.py:15-22: anisotropic RBF whose length scales cycle (lx, ly, lz) over the state index."""

    def __init__(self, lx, ly, lz, n_states):
        l = [lx, ly, lz]
        all_ls = np.zeros((1, n_states))
        for i in range(all_ls.shape[1]):
            all_ls[:, i] = l[i % 3]
        self.kernel = DeviceKernel(_lib.NK_KERNEL_RBF, all_ls)


class KernelWrapper:
    """regressors.py:24-26: Matern nu = 2.5."""

    def __init__(self, ls):
        self.kernel = DeviceKernel(_lib.NK_KERNEL_MATERN52, ls)


class LinearKernelWrapper:
    """regressors.py:28-30: DotProduct(sigma_0)."""

    def __init__(self, sigma):
        self.kernel = DeviceKernel(_lib.NK_KERNEL_LINEAR, None, sigma)
