#!/bin/bash
# up to how many tiles is a batch of the C3 code better dealt one team per tile over all XCDs than to one team per XCD?
L=gpurun_out/scatter_tiles.log
: > $L
for e in "LDPC_TEAM_SCATTER_TILES=4" "LDPC_TEAM_SCATTER_TILES=8" "LDPC_TEAM_SCATTER_TILES=2" "LDPC_TEAM_SCATTER_TILES=0"; do
  echo "== $e" >> $L
  env ${e//,/ } CASES=16384:0.10,16384:0.02,32768:0.10 VARIANT=4 BATCHES=64,128,192,256,320,384,512 timeout -k 10 200 python tools/smallbatch_probe.py 2>&1 | grep "^n " | cut -c1-120 >> $L
done
cat $L
