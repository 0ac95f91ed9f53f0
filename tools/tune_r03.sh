#!/bin/bash
# last tuning pass of round 3 on the three C3 workloads: static share of the dealing, mirrored order, running ahead
L=gpurun_out/tune_r03.log
: > $L
ENVS="LDPC_TEAM_DEBUG=0 LDPC_TEAM_STATIC=2 LDPC_TEAM_STATIC=4 LDPC_TEAM_FLIP=0 LDPC_TEAM_FLIP=1 LDPC_TEAM_FLIP=2 LDPC_TEAM_AHEAD=0 LDPC_TEAM_AHEAD_FROM=1 LDPC_TEAM_DEBUG=0" WLS="c3_full50 c3_waterfall c3_realistic" tools/bench_trio_ab.sh >> $L 2>&1
cat $L
