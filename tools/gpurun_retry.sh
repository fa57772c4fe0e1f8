#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy.   Usage: tools/gpurun_retry.sh <timeout s> '<command>'   (log: /tmp/gpu_try.log)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpu_try.log 2>&1
  grep -q "status=transient" /tmp/gpu_try.log || break
  sleep 100
done
