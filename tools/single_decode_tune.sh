#!/bin/bash
# a single decode! (and one tile) on the C3 code: team size, rows on chip, running ahead
L=gpurun_out/single_decode_tune.log
: > $L
for e in "LDPC_TEAM_DEBUG=0" "LDPC_TEAM_MAX=32" "LDPC_TEAM_MAX=96" "LDPC_TEAM_ROWS=0" "LDPC_TEAM_ROWS=0,LDPC_TEAM_MAX=96" "LDPC_TEAM_AHEAD=1" "LDPC_TEAM_AHEAD=1,LDPC_TEAM_AHEAD_FROM=1" "LDPC_TEAM_MIN_ROWS=256,LDPC_TEAM_MAX=96" "LDPC_TEAM_PER_CU=1"; do
  echo "== $e" >> $L
  env ${e//,/ } CASES=16384:0.02,16384:0.10 AUTO=1 BATCHES=1,64 timeout -k 10 100 python tools/smallbatch_probe.py 2>&1 | grep "^n " >> $L
done
cat $L
