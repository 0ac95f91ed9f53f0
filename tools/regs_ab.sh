#!/bin/bash
# rows in the top registers on / off, on 7 and on 8 XCDs: C3 full batch (bench.py), then the per-team phase times at 256 tiles
L=gpurun_out/regs_ab.log
: > $L
ENVS="${ENVS:-LDPC_TEAM_REGS=0 LDPC_TEAM_REGS=32 LDPC_TEAM_REGS=16 LDPC_TEAM_REGS=32,LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=400 LDPC_TEAM_REGS=32,LDPC_TEAM_STATIC=4 LDPC_TEAM_REGS=0 LDPC_TEAM_REGS=32}" WLS="${WLS:-c3_full50}" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_REGS=32" "LDPC_TEAM_REGS=32,LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=400"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
