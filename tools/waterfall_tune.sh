#!/bin/bash
# waterfall (per 0.06) and realistic (per 0.02) workloads of the C3 code by hand-off threshold and run-ahead lanes
L=gpurun_out/waterfall_tune.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_AHEAD=17 LDPC_TEAM_AHEAD=48 LDPC_DEFER_T0=12 LDPC_DEFER_T0=20 LDPC_DEFER_T0=24,LDPC_DEFER_T1=16 LDPC_DEFER_T0=16,LDPC_DEFER_T1=8 LDPC_TEAM_DEBUG=0" WLS="c3_waterfall c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
cat $L
