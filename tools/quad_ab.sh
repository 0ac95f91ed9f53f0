#!/bin/bash
# four bits at once in the variable sweep (LDPC_TEAM_PAIRS=3) against two (as built)
L=gpurun_out/quad_ab.log
: > $L
ENVS="${ENVS:-LDPC_TEAM_DEBUG=0 LDPC_TEAM_PAIRS=3 LDPC_TEAM_DEBUG=0 LDPC_TEAM_PAIRS=3}" WLS="c3_full50 c3_waterfall c3_realistic reg36_16380 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_PAIRS=1" "LDPC_TEAM_PAIRS=3"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
