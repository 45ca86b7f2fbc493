# PMC passes on the direct-difference kernel-matrix kernel (one counter group per run, program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in duffing cloth; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kmat_${w}_t -o run -- python3 $R/tools/kmat_bench.py $w 20 > $R/gpurun_out/kmat_${w}_t.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/kmat_${w}_p1 -o run -- python3 $R/tools/kmat_bench.py $w 3 > $R/gpurun_out/kmat_${w}_p1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/kmat_${w}_p2 -o run -- python3 $R/tools/kmat_bench.py $w 3 > $R/gpurun_out/kmat_${w}_p2.log 2>&1 || exit 1
done
find $R/gpurun_out -name "*kmat*" -maxdepth 1 | head; find $R/gpurun_out/kmat_duffing_t -type f | head
