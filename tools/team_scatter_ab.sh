#!/bin/bash
# persistent teams: members of a team inside one XCD (default) or dealt over all XCDs (LDPC_TEAM_SCATTER=1)
P=tools/team_mall_probe.py
L=gpurun_out/team_scatter.log
: > $L
for e in "X=0" "LDPC_TEAM_SCATTER=1" "LDPC_TEAM_SCATTER=1 LDPC_TEAM_DYNAMIC=0" "LDPC_TEAM_SCATTER=1 LDPC_TEAM_MAX=64 LDPC_TEAM_MIN_ROWS=1024 LDPC_TEAM_NO_MARGIN=1"; do
  echo "== $e" >> $L
  env $e BATCHES=${BATCHES:-512,4096} timeout -k 10 150 python $P >> $L 2>&1 || { echo "FAILED ($e)" >> $L; break; }
done
grep -v amdgpu.ids $L
