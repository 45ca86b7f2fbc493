cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_i1 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_i1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_i2 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_i2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_i3 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_i3.log 2>&1
ls $R/gpurun_out/pmc_i1 $R/gpurun_out/pmc_i2 $R/gpurun_out/pmc_i3
