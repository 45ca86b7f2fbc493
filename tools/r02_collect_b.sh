# round-2 evidence, part B: PMC passes (one counter group per run), callers' loops, soak, shape sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gram_1 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_gram_2 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_gram_3 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $O/pmc_gram_3.log 2>&1
for w in duffing duffing_rbf duffing_linear cloth; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kmat_${w}_t -o run -- python3 $R/tools/kmat_bench.py $w 20 > $O/kmat_${w}_t.log 2>&1
done
for w in duffing cloth; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/kmat_${w}_p1 -o run -- python3 $R/tools/kmat_bench.py $w 3 > $O/kmat_${w}_p1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/kmat_${w}_p2 -o run -- python3 $R/tools/kmat_bench.py $w 3 > $O/kmat_${w}_p2.log 2>&1
done
cd $R
python3 tools/lockstep_bench.py 16 64 32x2 30x3 > $O/lockstep_bench.log 2>&1
python3 tools/rollout_bench.py > $O/rollout_bench.log 2>&1
python3 tools/lift_latency.py >> $O/rollout_bench.log 2>&1
python3 tools/chain_mw_probe.py 500,6,6 1000,8,3 2000,8,6 > $O/chain_mw_probe.log 2>&1
NYSKOOP_CHAIN_MW=0 python3 tools/chain_mw_probe.py 500,6,6 2000,8,6 > $O/chain_mw_probe_stepwise.log 2>&1
NYSKOOP_HOST_PASSES=1 python3 tools/host_fit_bench.py > $O/host_fit_bench.log 2>&1
python3 tools/host_fit_bench.py >> $O/host_fit_bench.log 2>&1
python3 tools/cv_bench.py > $O/cv_bench.log 2>&1
python3 tools/jacobi_bench.py > $O/jacobi_bench.log 2>&1
NYSKOOP_PINV_BLOCK=0 python3 tools/jacobi_bench.py >> $O/jacobi_bench.log 2>&1
python3 tools/exact_kernel_bench.py 4000 > $O/exact_kernel_bench.log 2>&1
NYSKOOP_EXACT_HOST=1 python3 tools/exact_kernel_bench.py 4000 >> $O/exact_kernel_bench.log 2>&1
python3 tools/exact_kernel_bench.py 10000 >> $O/exact_kernel_bench.log 2>&1
NYSKOOP_EXACT_HOST=1 python3 tools/exact_kernel_bench.py 10000 >> $O/exact_kernel_bench.log 2>&1
python3 tools/kmat_epilogue_probe.py > $O/kmat_epilogue_probe.log 2>&1
python3 tools/rollout_fuzz.py 80 > $O/rollout_fuzz.log 2>&1
python3 tools/soak.py 30 > $O/soak.log 2>&1
python3 tools/shape_sweep.py 32 > $O/shape_sweep.log 2>&1
tail -4 $O/lockstep_bench.log | cut -c1-150; grep "^m=" $O/rollout_bench.log | cut -c1-300; grep PASSES $O/host_fit_bench.log; tail -3 $O/soak.log; tail -3 $O/shape_sweep.log
