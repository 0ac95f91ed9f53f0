#!/bin/bash
# the division of :147 in the variable sweep (default) against in the check sweep (libldpc_no_tform.so), alternating, same box
L=gpurun_out/tform_ab.log
: > $L
NT=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_no_tform.so
ENVS="LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$NT,LDPC_TEAM_DEBUG=0 LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$NT,LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=32 LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=400" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_STATIC=3" "LDPC_MI355X_EXP_LIB=$NT" "LDPC_TEAM_REGS=32"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
