#!/bin/bash
# where does the time of the rows-in-registers path go?  per-team phase times at 256 tiles by register rows per wave, and with the
# accessors compiled out (libldpc_fake_regs.so: wrong results, same code around them)
L=gpurun_out/regs_scaling.log
: > $L
for e in "LDPC_TEAM_REGS=0,LDPC_TEAM_STATIC=3" "LDPC_TEAM_REGS=8" "LDPC_TEAM_REGS=16" "LDPC_TEAM_REGS=32" "LDPC_TEAM_REGS=32,LDPC_TEAM_PAIRS=0" "LDPC_TEAM_REGS=32,LDPC_MI355X_EXP_LIB=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_fake_regs.so"; do
  echo "== $e" >> $L
  env ${e//,/ } LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "team rows|batch" >> $L
done
cat $L
