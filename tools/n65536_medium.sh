#!/bin/bash
# n = 65536 (128 MiB a slot) at medium batches, 50 iterations: what the plan does, eight persistent teams, the tile kernel
L=gpurun_out/n65536_medium.log
: > $L
for b in 1024 4096 16384; do
  for e in "LDPC_TEAM_DEBUG=1" "LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=4000"; do
    echo "== batch $b $e" >> $L
    env ${e//,/ } N=65536 VARIANT=0 LDPC_TEAM_DEBUG=1 BATCHES=$b timeout -k 10 200 python tools/team_mall_probe.py 2>&1 | grep -E "^batch|team kernel:" | cut -c1-210 >> $L
  done
  echo "== batch $b tile kernel" >> $L
  N=65536 VARIANT=1 BATCHES=$b timeout -k 10 200 python tools/team_mall_probe.py 2>&1 | grep -E "^batch" | cut -c1-120 >> $L
done
cat $L
