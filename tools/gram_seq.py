import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
X, Y, idx = O.make_c4()
Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(20., 20., 20., 384), gamma=1e-6, m=2000)
reg.nystrom_centers_output = np.ascontiguousarray(Y[idx].T)
out = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg.fit(Xd, Yd, fetch=False)
    out.append(((time.perf_counter() - t0) * 1e3, reg.fit_stats_["ms_gram_kernel_avg"], reg.fit_stats_["ms_kmat"]))
print("fit ms :", " ".join(f"{a:.1f}" for a, b, c in out))
print("gram ms:", " ".join(f"{b:.1f}" for a, b, c in out))
print("kmat ms:", " ".join(f"{c:.2f}" for a, b, c in out))
