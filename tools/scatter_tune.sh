#!/bin/bash
# one tile (a single decode!, 64 syndromes) and four tiles of the C3 code: members of the team that is dealt over all XCDs
L=gpurun_out/scatter_tune2.log
: > $L
for e in "LDPC_TEAM_SCATTER_MAX=96,LDPC_TEAM_SCATTER_ROWS=680" "LDPC_TEAM_SCATTER_MAX=128,LDPC_TEAM_SCATTER_ROWS=512" "LDPC_TEAM_SCATTER_MAX=192,LDPC_TEAM_SCATTER_ROWS=340" "LDPC_TEAM_SCATTER_MAX=256,LDPC_TEAM_SCATTER_ROWS=256" "LDPC_TEAM_SCATTER_MAX=128,LDPC_TEAM_SCATTER_ROWS=512,LDPC_TEAM_AHEAD=1"; do
  echo "== $e" >> $L
  env ${e//,/ } CASES=8192:0.10,16384:0.02,16384:0.10,32768:0.10 AUTO=1 BATCHES=1,64,128,256 timeout -k 10 170 python tools/smallbatch_probe.py 2>&1 | grep "^n " >> $L
done
cat $L
