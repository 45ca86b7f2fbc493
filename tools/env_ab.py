#!/usr/bin/env python3
"""Same-process A/B of a switch the library reads per fit / per launch (e.g. NYSKOOP_CHOL_FIX, NYSKOOP_TN_ASM) on the headline
fit: alternating, several rounds, median of the fits of each round.  MI355X boxes differ by several per cent, so only numbers
from one process compare.  Usage: env_ab.py VAR value [value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
var, vals = sys.argv[1], sys.argv[2:]
X, Y, idx = O.make_c4()
Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(20., 20., 20., 384), gamma=1e-6, m=2000)
reg.nystrom_centers_output = np.ascontiguousarray(Y[idx].T)
ref = None
for rnd in range(4):
    for v in vals:
        os.environ[var] = v
        ts = []
        for i in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reg.fit(Xd, Yd, fetch=False)
            ts.append((time.perf_counter() - t0) * 1e3)
        st = reg.fit_stats_
        A = np.array(reg.A)
        if ref is None: ref = A
        print(f"round {rnd} {var}={v}: fit {np.median(ts[1:]):.2f} ms (min {min(ts[1:]):.2f})  kmat {st['ms_kmat']:.2f} gram {st['ms_gram']:.2f} "
              f"launch {st['ms_gram_kernel_avg']:.2f} sqrt {st['ms_sqrt']:.2f} solve {st['ms_solve']:.2f} total {st['ms_total']:.2f} | A vs first "
              f"{np.linalg.norm(A - ref) / np.linalg.norm(ref):.1e}", flush=True)
