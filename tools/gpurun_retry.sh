#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged): tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
