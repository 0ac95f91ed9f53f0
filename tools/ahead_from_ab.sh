#!/bin/bash
L=gpurun_out/ahead_from_ab.log
: > $L
ENVS="LDPC_TEAM_AHEAD_FROM=2 LDPC_TEAM_AHEAD=0 LDPC_TEAM_AHEAD_FROM=2 LDPC_TEAM_AHEAD=0" WLS="c3_realistic c3_waterfall c3_full50" tools/bench_trio_ab.sh >> $L 2>&1
cat $L
