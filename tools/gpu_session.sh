#!/bin/bash
# Helper for one gpurun call: runs the given steps in order, each under its own timeout, logging to
# gpurun_out/<tag>_<k>.log; stops at the first step that TIMES OUT (a hung GPU step must not be followed by
# another one), carries on after an ordinary failure.  Usage: tools/gpu_session.sh <tag> <timeout_s> '<cmd>' [<timeout_s> '<cmd>' ...]
TAG=$1; shift
mkdir -p gpurun_out
k=0
while [ $# -ge 2 ]; do
    T=$1; CMD=$2; shift 2; k=$((k+1))
    echo "[$TAG step $k] $CMD"
    timeout -k 10 $T bash -c "$CMD" > gpurun_out/${TAG}_$k.log 2>&1
    rc=$?
    echo "[$TAG step $k] exit $rc"; tail -3 gpurun_out/${TAG}_$k.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$TAG] step $k timed out: stopping"; exit 1; fi
    if grep -q "Memory access fault" gpurun_out/${TAG}_$k.log; then echo "[$TAG] step $k faulted on the GPU: stopping"; exit 1; fi
done
exit 0
