"""Where does the host-side time of one fit go?  (run on the GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
from bench import make_c4
n, m, d, p = 100000, 2000, 384, 6
X, Y, idx = make_c4(n, d, p, m)
dev = torch.device("cuda", 0)
Xd, Yd = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
Z = np.ascontiguousarray(Y[idx])
ctx = nk.get_context(0)
def fit():
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-6, m=m)
    reg.nystrom_centers_output = Z.T
    reg.fit(Xd, Yd)
    return reg
fit(); fit()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); regs = [fit() for _ in range(3)]; t1 = time.perf_counter()
pr.disable()
print("wall per fit %.2f ms; device %.2f ms" % ((t1 - t0) / 3 * 1e3, regs[-1].fit_stats_["ms_total"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
