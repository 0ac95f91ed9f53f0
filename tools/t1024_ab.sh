#!/bin/bash
# members of 16 waves (libldpc_t1024.so: -DLDPC_TEAM_THREADS=1024) against 8, same box
L=gpurun_out/t1024_ab.log
: > $L
V=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_t1024.so
ENVS="LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$V,LDPC_TEAM_DEBUG=0 LDPC_MI355X_EXP_LIB=$V,LDPC_TEAM_STATIC=0 LDPC_MI355X_EXP_LIB=$V,LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=400 LDPC_TEAM_DEBUG=0" WLS="c3_full50 c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
for e in "LDPC_MI355X_EXP_LIB=$V"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
