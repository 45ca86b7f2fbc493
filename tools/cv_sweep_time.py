#!/usr/bin/env python3
"""The synthetic 405-unit CV sweep of bench.py (cv_sweep_rate, lock-step batched) alone -- for A/B runs of environment switches
and for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nys_koop_lqr_amd as nk
import bench
bench.real_cloth_grid_rate = lambda *a: None
r = bench.cv_sweep_rate(nk)
print({k: r[k] for k in ("units_per_s", "seconds", "seconds_min_max", "units_per_s_unbatched", "bit_identical_to_unbatched")}, flush=True)
