# round-2 evidence, part A: the GPU test suite (with its printed measurements) and the bench line, plain and profiled
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
python3 bench.py > $O/bench_full.log 2>&1; echo "bench rc $?" >> $O/bench_full.log
grep '^{"metric"' $O/bench_full.log > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o run -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-baseline none --no-extras > $O/bench_profiled.log 2>&1; echo "profiled rc $?" >> $O/bench_profiled.log
grep '^{"metric"' $O/bench_profiled.log > $O/bench_line_profiled.json
tail -3 $O/gpu_tests.log; tail -2 $O/bench_full.log | cut -c1-300; tail -1 $O/bench_profiled.log
