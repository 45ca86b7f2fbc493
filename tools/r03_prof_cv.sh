#!/bin/bash
# synthetic CV sweep: timing with / without the correction step (alternating), then kernel statistics of both
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cvprof; mkdir -p $O
cd $R
for rep in 1 2; do for f in 1 0 2; do
  NYSKOOP_CHOL_FIX=$f timeout -k 10 200 python3 tools/cv_sweep_time.py > $O/t$f.log 2>&1 || { echo "run $f failed"; tail -5 $O/t$f.log; exit 1; }
  echo "fix=$f $(tail -1 $O/t$f.log | cut -c1-200)"
done; done
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  export NYSKOOP_CHOL_FIX=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f$f -o cv --output-format csv -- python3 $R/tools/cv_sweep_time.py > $O/run$f.log 2>&1 || { echo "prof $f failed"; tail -5 $O/run$f.log; exit 1; }
  echo "== fix=$f"; find $O/f$f -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-160'
done
