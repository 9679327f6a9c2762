#!/bin/bash
# tools/gpu_retry.sh <timeout_s> '<command>': tools/gpu.sh, retried while gpurun reports "no box or slot free" (exit 3: nothing charged)
cd "$(dirname "$0")/.."
for k in 1 2 3 4 5 6 7 8; do
  bash tools/gpu.sh "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 100
done
exit 3
