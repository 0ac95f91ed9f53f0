#!/bin/bash
# large (4,8)-regular codes at the FULL batch (65,536 syndromes x 50 iterations): eight persistent teams (slots far beyond
# the Infinity Cache) against the tile kernel (768 slots, HBM streaming)
L=gpurun_out/midsize_plan2.log
: > $L
for n in 40960 49152 65536; do
  echo "== n $n eight teams" >> $L
  LDPC_TEAM_XCDS=8 LDPC_TEAM_CACHE_MIB=4000 N=$n BATCHES=65536 timeout -k 10 280 python tools/team_mall_probe.py 2>&1 | grep -E "^batch" | cut -c1-200 >> $L
  echo "== n $n tile kernel" >> $L
  N=$n VARIANT=1 BATCHES=65536 timeout -k 10 280 python tools/team_mall_probe.py 2>&1 | grep -E "^batch" | cut -c1-120 >> $L
done
cat $L
