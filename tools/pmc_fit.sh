# PMC passes over a few whole fits (bench.py, no CPU legs): per-kernel MFMA busy, VALU instruction mix, waits
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_fit_1 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-extras > $O/pmc_fit_1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_fit_2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-extras > $O/pmc_fit_2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_fit_3 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-extras > $O/pmc_fit_3.log 2>&1
cd $R; python3 - <<PY
import csv
from collections import defaultdict
for i in (1,2,3):
    try: rows=list(csv.DictReader(open(f"gpurun_out/r02/pmc_fit_{i}/run_counter_collection.csv")))
    except Exception as e: print(i, e); continue
    agg=defaultdict(lambda: defaultdict(list))
    for r in rows:
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        if "gemm_tn_f64_kernel<1>" in k or "gram_fused" in k or "kmat_kernel" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(i, k, {c: sum(x)/len(x) for c,x in v.items()}, "n", len(next(iter(v.values()))))
PY
