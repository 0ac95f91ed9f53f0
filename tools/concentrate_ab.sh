#!/bin/bash
# on-chip rows gathered into whole checks (LDPC_TEAM_CONCENTRATE=1) and the mirrored order of the upper waves
# (LDPC_TEAM_FLIP) against the build's default; C3 workloads, the other regular codes, per-team phase times
L=gpurun_out/concentrate_ab.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_CONCENTRATE=1 LDPC_TEAM_CONCENTRATE=1,LDPC_TEAM_FLIP=1 LDPC_TEAM_CONCENTRATE=1,LDPC_TEAM_FLIP=3 LDPC_TEAM_FLIP=3 LDPC_TEAM_DEBUG=0 LDPC_TEAM_CONCENTRATE=1" WLS="c3_full50 c3_waterfall c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_CONCENTRATE=1 LDPC_TEAM_CONCENTRATE=1,LDPC_TEAM_FLIP=3 LDPC_TEAM_REGS=0" WLS="reg36_16380 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_CONCENTRATE=0" "LDPC_TEAM_CONCENTRATE=1" "LDPC_TEAM_CONCENTRATE=1,LDPC_TEAM_FLIP=3"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
  env ${e//,/ } WR=6 WC=3 N=16380 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
