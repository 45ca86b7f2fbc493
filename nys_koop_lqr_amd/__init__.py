"""nys_koop_lqr_amd -- MI355X (gfx950) drop-in for the Nystrom-Koopman fit / lift / predict / rollout surface of
LCSL/nys-koop-lqr's regressors.py.  Python host classes call hand-written HIP kernels through the C-ABI of
libnyskoop.so (include/nyskoop.h) with ctypes.  There is no CPU fallback: importing works anywhere, but any
compute call without the built library and a gfx950 device raises.
"""
from .kernels import KernelWrapper, LinearKernelWrapper, ThreeDimensionalKernel  # noqa: F401
from .regressors import KoopmanKernelRegressor, KoopmanNystromRegressor, KoopmanRegressor, linear_rollout  # noqa: F401
from ._lib import NyskoopError, get_context, library_path, shutdown  # noqa: F401

__all__ = [
    "KoopmanRegressor", "KoopmanNystromRegressor", "KoopmanKernelRegressor", "ThreeDimensionalKernel", "KernelWrapper",
    "LinearKernelWrapper", "NyskoopError", "get_context", "library_path", "linear_rollout", "shutdown",
]
