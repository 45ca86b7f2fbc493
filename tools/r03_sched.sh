#!/bin/bash
# schedule experiments on the headline fit (same box): CU-masked side stream x look-ahead Cholesky
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r03_sched.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stages_ms']
print('ms/fit %.2f | total %.2f kmat %.2f gram %.2f sqrt %.2f solve %.2f' % (j['ms_per_step'], s['ms_total'], s['ms_kmat'], s['ms_gram'], s['ms_sqrt'], s['ms_solve']))" >> $L 2>&1; }
run A=base
run NYSKOOP_SQRT_AFTER_CHAIN=1
run NYSKOOP_SQRT_AFTER_CHAIN=1 NYSKOOP_CHOL_LOOKAHEAD=1
run NYSKOOP_SQRT_AFTER_CHAIN=1 NYSKOOP_CHOL_LOOKAHEAD=1 NYSKOOP_CHOL_LOOKAHEAD_PREP=1
run NYSKOOP_CHOL_LOOKAHEAD=1
run A=base
cat $L
