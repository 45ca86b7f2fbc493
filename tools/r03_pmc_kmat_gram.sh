#!/bin/bash
# PMC pass on the headline fit: matrix-pipe occupancy and wave stalls of the Gram-form kernel blocks (gemm_tn_f64_kernel<1>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_kg; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -o run -- python3 $R/tools/fit_loop.py 4 > $O/p1.log 2>&1 || { echo "pmc pass failed"; tail -5 $O/p1.log; exit 1; }
python3 - <<PY
import csv, collections
by = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}; name = {}
for r in csv.DictReader(open("$O/p1/run_counter_collection.csv")):
    k = int(r["Dispatch_Id"]); by[k][r["Counter_Name"]] += float(r["Counter_Value"]); name[k] = r["Kernel_Name"]
    dur[k] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
for pat in ("gemm_tn_f64_kernel<1>", "gram_fused"):
    ks = [k for k in sorted(by) if pat in name[k]][2:]
    if not ks: continue
    avg = lambda c: sum(by[k][c] for k in ks) / len(ks)
    gui = avg("GRBM_GUI_ACTIVE")
    print(pat, "launches", len(ks), "ms %.3f" % (sum(dur[k] for k in ks) / len(ks)), "MFMA busy %.3f" % (avg("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024 / 8 * 8) if gui else 0),
          "raw", {c: "%.3e" % avg(c) for c in by[ks[0]]})
PY
