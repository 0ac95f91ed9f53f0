#!/bin/bash
# barrier + test share: the test's words ORed by every wave directly (as built) against gathered in LDS
# (libldpc_test_lds.so); poll sleep 2 instead of 16 (libldpc_sleep2.so).  Alternating, same box.
L=gpurun_out/barrier_ab.log
: > $L
C=$PWD/ldpcdecoders.jl_amd/csrc
ENVS="LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_test_lds.so,LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_sleep2.so,LDPC_TEAM_DEBUG=0 LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_test_lds.so,LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$C/libldpc_sleep2.so,LDPC_TEAM_DEBUG=0" WLS="c3_full50 c3_waterfall c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_TEAM_DEBUG=1" "LDPC_MI355X_EXP_LIB=$C/libldpc_test_lds.so" "LDPC_MI355X_EXP_LIB=$C/libldpc_sleep2.so"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "batch" >> $L
done
cat $L
