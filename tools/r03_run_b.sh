#!/bin/bash
# round-3 GPU batch B: full GPU suite, headline bench, MFMA-busy PMC pass on the Gram launch, per-unit cloth grid table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # step <log> <cmd...>
  local log=$1; shift
  "$@" > gpurun_out/$log 2>&1
  local rc=$?
  if grep -q "Memory access fault\|HSA_STATUS_ERROR" gpurun_out/$log; then echo "GPU fault in $log"; tail -5 gpurun_out/$log; exit 70; fi
  if [ $rc -ne 0 ]; then echo "step $log failed rc=$rc"; tail -60 gpurun_out/$log; exit $rc; fi
}
step b_tn.log timeout -k 10 300 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu && tail -2 gpurun_out/b_tn.log
# (the whole suite: failures are reported, the batch goes on -- a GPU fault still stops it)
timeout -k 10 900 python3 -m pytest tests -q -m gpu -s > gpurun_out/b_tests.log 2>&1; tail -8 gpurun_out/b_tests.log
if grep -q "Memory access fault" gpurun_out/b_tests.log; then echo "GPU fault in the suite"; exit 70; fi
step b_bench.log timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras && python3 -c "
import json,sys
j=json.loads(open('gpurun_out/b_bench.log').read().strip().splitlines()[-1]); print('bench', j['ms_per_step'], j['stages_ms'], j['roofline']['frac'])"
NYSKOOP_CHOL_LOOKAHEAD=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/b_bench_nola.log 2>&1; python3 -c "
import json,sys
j=json.loads(open('gpurun_out/b_bench_nola.log').read().strip().splitlines()[-1]); print('bench no look-ahead', j['ms_per_step'], j['stages_ms'])"
step b_units.log timeout -k 10 300 python3 tools/cloth_grid_units.py gpurun_out/r03_cloth_units.txt && tail -6 gpurun_out/r03_cloth_units.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_b1 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_b1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_b1.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(f'{R}/gpurun_out/pmc_b1/**/*counter_collection.csv', recursive=True):
    acc={}
    for r in csv.DictReader(open(f)):
        if 'gram_fused' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
            acc.setdefault('dur_ms',[]).append((float(r['End_Timestamp'])-float(r['Start_Timestamp']))*1e-6)
    for k,v in acc.items(): print(k,['%.5g'%x for x in v[:4]])
    g=acc['GRBM_GUI_ACTIVE'][-1]; print('mfma busy', acc['SQ_VALU_MFMA_BUSY_CYCLES'][-1]/(1024*g/8), 'clock GHz', g/8/ (acc['dur_ms'][-1]*1e-3)/1e9)
PY
