#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 tools/f32_diag.py > gpurun_out/r03_f32_diag.log 2>&1; rc=$?
cat gpurun_out/r03_f32_diag.log | cut -c1-2500
exit $rc
