#!/bin/bash
# gpurun with patience: exit code 3 (no box / slot free, nothing charged) is retried every 90 s, up to 40 times.
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
