#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 tools/tn_ab.py > gpurun_out/r03_tn_ab.log 2>&1; rc=$?
grep "round" gpurun_out/r03_tn_ab.log || tail -20 gpurun_out/r03_tn_ab.log
exit $rc
