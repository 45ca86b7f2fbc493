#!/usr/bin/env python3
"""Same-box A/B of the TN engine's hand-scheduled k steps against the compiler-scheduled form (NYSKOOP_TN_ASM=0/1, read per
launch): the fused Gram launch alone and whole fits, alternating, several rounds.  MI355X boxes differ by several per cent
in sustained fp64 MFMA rate (power management), so only numbers from ONE process on ONE box compare."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nys_koop_lqr_amd import _lib
import nys_koop_lqr_amd as nk

ctx = _lib.get_context(0)
def gram(reps=4):
    ms, fl = C.c_double(), C.c_double()
    _lib.check(ctx.lib.nk_bench_gram(ctx.handle, 100000, 2000, 6, 384, reps, C.byref(ms), C.byref(fl)))
    return ms.value, fl.value / ms.value * 1e-9
import torch
from oracle import nk_oracle as O
X, Y, idx = O.make_c4()
Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
def fits(k=6):
    reg = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(20., 20., 20., 384), gamma=1e-6, m=2000)
    reg.nystrom_centers_output = np.ascontiguousarray(Y[idx].T)
    ts, st = [], None
    for i in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reg.fit(Xd, Yd, fetch=False)
        ts.append((time.perf_counter() - t0) * 1e3); st = reg.fit_stats_
    return np.median(ts[1:]), st
for rnd in range(3):
    for mode in ("1", "0"):
        os.environ["NYSKOOP_TN_ASM"] = mode
        g = gram()
        f, st = fits()
        print(f"round {rnd} asm={mode}: gram {g[0]:.3f} ms ({g[1]:.1f} TF) | fit {f:.2f} ms  kmat {st['ms_kmat']:.2f} gram {st['ms_gram']:.2f} "
              f"sqrt {st['ms_sqrt']:.2f} solve {st['ms_solve']:.2f} total {st['ms_total']:.2f}", flush=True)
