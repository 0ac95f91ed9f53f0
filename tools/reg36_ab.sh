#!/bin/bash
# (3,6)-regular n = 16380 (eight whole slots fit the cache anyway): register rows on / off with whole checks on chip
L=gpurun_out/reg36_ab.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0 LDPC_TEAM_REGS=16 LDPC_TEAM_REGS=0,LDPC_TEAM_FLIP=0 LDPC_TEAM_REGS=0,LDPC_TEAM_STATIC=0 LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0" WLS="reg36_16380" tools/bench_trio_ab.sh >> $L 2>&1
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0 LDPC_TEAM_REGS=0,LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=400" WLS="c3_full50 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_REGS=32" "LDPC_TEAM_REGS=0"; do
  echo "== $e" >> $L
  env ${e//,/ } WR=6 WC=3 N=16380 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
