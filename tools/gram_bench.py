#!/usr/bin/env python3
"""Time (or profile with rocprofv3 --pmc) the fused Gram launch alone: python3 tools/gram_bench.py [n m p d reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nys_koop_lqr_amd import _lib
n, m, p, d, reps = [int(x) for x in (sys.argv[1:6] + ["100000", "2000", "6", "384", "3"][len(sys.argv) - 1:])]
ctx = _lib.get_context(0)
ms, fl = C.c_double(), C.c_double()
_lib.check(ctx.lib.nk_bench_gram(ctx.handle, n, m, p, d, reps, C.byref(ms), C.byref(fl)))
print(f"gram n={n} m={m} p={p} d={d}: {ms.value:.3f} ms/launch, {fl.value / ms.value * 1e-9:.2f} TFLOP/s (algorithmic, SYRK-aware)")
