#!/bin/bash
# round-3 quick check on the GPU box: TN-engine tests first (a wrong kernel must not reach the long runs), then the Gram launch
# alone and the headline bench.  A step that fails (or whose log shows a GPU fault) stops the script with a non-zero code.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # step <log> <cmd...>
  local log=$1; shift
  "$@" > gpurun_out/$log 2>&1
  local rc=$?
  if grep -q "Memory access fault\|HSA_STATUS_ERROR" gpurun_out/$log; then echo "GPU fault in $log"; tail -5 gpurun_out/$log; exit 70; fi
  if [ $rc -ne 0 ]; then echo "step $log failed rc=$rc"; tail -40 gpurun_out/$log; exit $rc; fi
}
step q_tn.log timeout -k 10 300 python3 -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "tn_engine" && tail -3 gpurun_out/q_tn.log
step q_gram.log timeout -k 10 120 python3 tools/gram_bench.py 100000 2000 6 384 5 && cat gpurun_out/q_gram.log
step q_tests.log timeout -k 10 900 python3 -m pytest tests -x -q -m gpu ${QUICK_K:+-k "$QUICK_K"} && tail -3 gpurun_out/q_tests.log
step q_bench.log timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras && tail -2 gpurun_out/q_bench.log
