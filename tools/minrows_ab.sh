#!/bin/bash
# teams of 23 (2048 message rows a member per sweep at least) against teams of 32 on the (3,6) n = 16380 code; smaller codes too
L=gpurun_out/minrows_ab.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_MIN_ROWS=1500 LDPC_TEAM_MIN_ROWS=1024 LDPC_TEAM_DEBUG=0 LDPC_TEAM_MIN_ROWS=1500" WLS="reg36_16380" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_DEBUG=1" "LDPC_TEAM_MIN_ROWS=1500" "LDPC_TEAM_MIN_ROWS=1024"; do
  echo "== $e" >> $L
  env ${e//,/ } WR=6 WC=3 N=16380 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
  env ${e//,/ } WR=8 WC=4 N=12288 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
  env ${e//,/ } WR=8 WC=4 N=8192 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
