#!/bin/bash
# kernel statistics of the headline fit with and without the correction step of the diagonal-block solves
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fixprof; mkdir -p $O
for f in 1 0; do
  export NYSKOOP_CHOL_FIX=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f$f -o fit --output-format csv -- python3 $R/tools/fit_loop.py 8 > $O/run$f.log 2>&1 || { echo "prof $f failed"; tail -5 $O/run$f.log; exit 1; }
  echo "== fix=$f"; grep "ms per fit" $O/run$f.log | cut -c1-120
  find $O/f$f -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -1 {}; grep -i "panel\|trsm\|potrf\|trail" {}'
done
