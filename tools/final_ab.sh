#!/bin/bash
# default build (rows in registers, eight teams) against registers off (seven teams), three workloads + other codes
L=gpurun_out/final_ab.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0 LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0 LDPC_TEAM_STATIC=2 LDPC_TEAM_XCDS=7" WLS="c3_full50 c3_waterfall c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0 LDPC_TEAM_DEBUG=0 LDPC_TEAM_REGS=0" WLS="reg36_16380 wide_16000_10_5" tools/bench_trio_ab.sh >> $L 2>&1
cat $L
