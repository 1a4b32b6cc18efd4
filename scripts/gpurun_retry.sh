#!/bin/bash
# gpurun_retry.sh <timeout> <log> <command...>: submits through gpurun and retries while the pod's GPU slots are busy (exit code 3 / "transient": nothing is charged)
T=$1; LOG=$2; shift 2
for attempt in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if grep -q "status=transient" $LOG || [ $rc -eq 3 ]; then sleep 60; continue; fi
  exit $rc
done
exit 3
