#!/usr/bin/env python3
"""The real 405-unit cloth grid (bench.py's real_cloth_grid_rate) alone: seconds per sweep, worst score error, and the library's
slow-path counters -- for A/B runs of environment switches and for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
import bench
r = bench.real_cloth_grid_rate(nk, 32, 2)
print({k: r[k] for k in ("real_grid_seconds", "real_grid_seconds_min_max", "real_grid_max_rel_score_error",
                         "real_grid_best_index_matches_reference")}, _lib.runtime_counters(), flush=True)
