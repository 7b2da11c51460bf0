#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> '<command>' -- retries gpurun while the pod has no free slot / box (exit 3: nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
