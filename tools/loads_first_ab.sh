#!/bin/bash
# the row loads of a node group kept in front of the arithmetic (scheduling barrier; as built) against the scheduler's own
# order (libldpc_loads_sunk.so: -DLDPC_LOADS_FIRST=0).  Alternating, same box.
L=gpurun_out/loads_first_ab.log
: > $L
C=$PWD/ldpcdecoders.jl_amd/csrc
ENVS="LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_loads_sunk.so,LDPC_TEAM_DEBUG=0 LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_loads_sunk.so,LDPC_TEAM_DEBUG=0" WLS="c3_full50 c3_waterfall c3_realistic reg36_16380 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_DEBUG=1" "LDPC_MI355X_EXP_LIB=$C/libldpc_loads_sunk.so"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "batch" >> $L
done
sed -i "s#$C/##g" $L
cat $L
