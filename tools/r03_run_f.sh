#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NYSKOOP_CV_TRACE=1 timeout -k 10 600 python3 - > gpurun_out/r03_f_grid.log 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench, nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
nk.get_context()
for shape in ((32, 2), (64, 1), (32, 3)):
    print(shape, json.dumps(bench.real_cloth_grid_rate(nk, *shape)), flush=True)
print(_lib.runtime_counters())
PY
cat gpurun_out/r03_f_grid.log | cut -c1-400
