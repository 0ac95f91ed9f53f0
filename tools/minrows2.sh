#!/bin/bash
# the one team of an XCD below 1500 rows a member: n = 9216 / 10240 / 11264 (4,8) and n = 12288 (3,6)
L=gpurun_out/minrows2.log
: > $L
for c in "8 4 9216" "8 4 10240" "8 4 11264" "6 3 12288"; do
  set -- $c
  for e in "LDPC_TEAM_DEBUG=1" "LDPC_TEAM_MIN_ROWS=1100" "LDPC_TEAM_MIN_ROWS=2048"; do
    echo "== ($1,$2) n $3 $e" >> $L
    env ${e//,/ } WR=$1 WC=$2 N=$3 LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 100 python tools/team_mall_probe.py 2>&1 | grep -E "^batch" | cut -c1-215 >> $L
  done
done
cat $L
