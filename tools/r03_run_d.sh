#!/bin/bash
# round-3 GPU batch D: the whole GPU suite with printed measurements, then the bench line with its extras (no CPU baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -s -p no:cacheprovider > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_gpu_tests.log
tail -12 gpurun_out/r03_gpu_tests.log
if grep -q "Memory access fault" gpurun_out/r03_gpu_tests.log; then echo "GPU fault in the suite"; exit 70; fi
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --cpu-baseline none > gpurun_out/r03_bench_extras.log 2>&1; echo "bench rc $?"
tail -1 gpurun_out/r03_bench_extras.log | cut -c1-3000
