#!/bin/bash
# round-3 evidence, part B: profiled bench (kernel trace + stats), PMC passes on the fused Gram launch (one counter group per run)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o run -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --no-extras > $O/bench_profiled.log 2>&1; echo "profiled rc $?" >> $O/bench_profiled.log
grep '^{"metric"' $O/bench_profiled.log > $O/bench_line_profiled.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gram_1 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_gram_2 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_gram_3 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_3.log 2>&1
cd $R
python3 -m pytest tests/test_gpu_configs.py -m gpu -q -s -p no:cacheprovider -k "cloth_cv" > $O/gpu_tests_cloth.log 2>&1; tail -2 $O/gpu_tests_cloth.log
python3 tools/cloth_grid_units.py $O/cloth_units.txt > $O/cloth_units.log 2>&1; tail -6 $O/cloth_units.txt | cut -c1-400
python3 tools/gram_bench.py 100000 2000 6 384 5 > $O/gram_bench.log 2>&1; cat $O/gram_bench.log | tail -1
python3 tools/lockstep_bench.py 16 64 32x2 30x3 > $O/lockstep_bench.log 2>&1; tail -3 $O/lockstep_bench.log
tail -1 $O/bench_profiled.log | cut -c1-200; ls $O
