#!/bin/bash
# Gram-launch experiments (round 3): where do the idle matrix-pipe cycles come from?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/gram_exp.log
: > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 120 python3 tools/gram_bench.py 100000 2000 6 384 4 >> $L 2>&1 || { echo "FAILED" >> $L; tail -5 $L; exit 1; }; if grep -q "Memory access fault" $L; then echo FAULT; exit 70; fi; }
run A=default
run NYSKOOP_TN_KMASK=0x3ff
run NYSKOOP_TN_KMASK=0xf
run NYSKOOP_TN_SPLITK=16
run NYSKOOP_TN_SPLITK=4
run NYSKOOP_TN_SPLITK=24
grep "==\|gram n" $L
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq1 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_sq1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_sq2 -o run -- python3 $R/tools/gram_bench.py 100000 2000 6 384 3 > $R/gpurun_out/pmc_sq2.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq2.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
for d in ('pmc_sq1','pmc_sq2'):
    for f in glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            if 'gram_fused' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,['%.4g'%x for x in v])
PY
