#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -q -m gpu -s -p no:cacheprovider -k "rank_deficient or pinv or cloth or sweep or lockstep or counters or duffing or fp32 or jacobi or solve_spd" > gpurun_out/r03_e_tests.log 2>&1; echo "pytest rc $?"
tail -8 gpurun_out/r03_e_tests.log
if grep -q "Memory access fault" gpurun_out/r03_e_tests.log; then exit 70; fi
NYSKOOP_CV_TRACE=1 timeout -k 10 300 python3 tools/cloth_grid_units.py gpurun_out/r03_cloth_units.txt > gpurun_out/r03_e_units.log 2>&1; tail -5 gpurun_out/r03_cloth_units.txt
timeout -k 10 600 python3 - > gpurun_out/r03_e_grid.log 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench, nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
nk.get_context()
print(json.dumps(bench.real_cloth_grid_rate(nk, 32, 2)))
print(_lib.runtime_counters())
PY
tail -3 gpurun_out/r03_e_grid.log
