#!/bin/bash
# round-3 evidence, part A: the GPU test suite (with its printed measurements), smoke, per-unit cloth table, the bench line (full protocol)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
if grep -q "Memory access fault" $O/gpu_tests.log; then echo "GPU fault"; exit 70; fi
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-400
python3 tools/cloth_grid_units.py $O/cloth_units.txt > $O/cloth_units.log 2>&1; tail -5 $O/cloth_units.txt
python3 bench.py > $O/bench_full.log 2>&1; echo "bench rc $?" >> $O/bench_full.log
grep '^{"metric"' $O/bench_full.log > $O/bench_line.json; tail -2 $O/bench_full.log | cut -c1-600
