#!/bin/bash
# usage: tools/gpu.sh <timeout_s> '<command>'   -- gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
