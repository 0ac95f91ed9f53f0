#!/bin/bash
# whole-check fast paths (registers / LDS) and the paired variable update with the first edge on chip: as built, with only
# whole checks kept on chip (LDPC_TEAM_CONCENTRATE=2), without register rows
L=gpurun_out/fastpath_ab.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_CONCENTRATE=2 LDPC_TEAM_REGS=0 LDPC_TEAM_REGS=0,LDPC_TEAM_CONCENTRATE=2 LDPC_TEAM_DEBUG=0 LDPC_TEAM_CONCENTRATE=2" WLS="c3_full50 c3_waterfall c3_realistic reg36_16380 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_CONCENTRATE=1" "LDPC_TEAM_CONCENTRATE=2"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
  env ${e//,/ } WR=6 WC=3 N=16380 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
