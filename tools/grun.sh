#!/bin/bash
# gpurun wrapper: retries only when no GPU slot/box is free (exit code 3), never on a failing command
t=${GRUN_TIMEOUT:-900}
for i in 1 2 3 4 5 6; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
