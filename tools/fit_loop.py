#!/usr/bin/env python3
"""The headline fit in a loop (HBM-resident inputs, no fetch) -- the program to put behind rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X, Y, idx = O.make_c4()
Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(20., 20., 20., 384), gamma=1e-6, m=2000)
reg.nystrom_centers_output = np.ascontiguousarray(Y[idx].T)
ts = []
for i in range(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg.fit(Xd, Yd, fetch=False)
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per fit:", np.round(ts, 2), reg.fit_stats_)
