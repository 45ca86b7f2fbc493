"""Time of the Jacobi-SVD fallback alone (nk_solve_spd with NYSKOOP_FORCE_PINV=1) on an ill-conditioned 506 x 506 system
of the cloth shape: per sweep, single-launch sweeps against one launch per round (NYSKOOP_PINV_SWEEP_LAUNCH=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NYSKOOP_FORCE_PINV"] = "1"
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd.regressors import KoopmanKernelRegressor
rng = np.random.default_rng(0)
m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 506, 808
# a Gram matrix with a spectrum that decays through the rounding level, like the gamma = 1e-7 cloth candidates
Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
s = np.logspace(0, -17, m)
P = (Q * s) @ Q.T
P = (P + P.T) / 2
R = rng.standard_normal((m, 8))
ctx = nk.get_context()
for rep in range(3):
    t0 = time.perf_counter()
    X = KoopmanKernelRegressor._solve_spd(ctx, P, R)
    dt = time.perf_counter() - t0
    print(f"m={m}: solve through the Jacobi fallback {dt * 1e3:.1f} ms (NYSKOOP_PINV_SWEEP_LAUNCH={os.environ.get('NYSKOOP_PINV_SWEEP_LAUNCH', 'auto: a launch per round outside a lock-step group')})", flush=True)
print("residual of the kept part:", np.linalg.norm(P @ X - R) / np.linalg.norm(R))
