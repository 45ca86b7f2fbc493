#!/bin/bash
# round-3 GPU batch C: C5 stress with both engines, kernel timeline of the headline fit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 tools/stress_c5.py > gpurun_out/r03_stress_c5.log 2>&1; rc=$?
cat gpurun_out/r03_stress_c5.log | tail -12
if grep -q "Memory access fault" gpurun_out/r03_stress_c5.log; then exit 70; fi
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -o run -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-baseline none --no-extras > $R/gpurun_out/r03_bench_profiled.log 2>&1 || { tail -5 $R/gpurun_out/r03_bench_profiled.log; exit 1; }
cd $R
python3 tools/timeline.py $(find gpurun_out/r03_prof -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/r03_fit_timeline.txt 2>&1
wc -l gpurun_out/r03_fit_timeline.txt; grep -n "gram_fused" gpurun_out/r03_fit_timeline.txt | head -3
