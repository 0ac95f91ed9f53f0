#!/bin/bash
# persistent teams in the Infinity Cache: how the chunks of a sweep are dealt (LDPC_TEAM_DYNAMIC 0 static, 1 per member
# from LDS, 2 per team from the XCD's L2), against round 1's one-team-per-tile geometry (LDPC_TEAM_CACHE_MIB=0):
# C3 code, all 50 iterations, by batch size
P=tools/team_mall_probe.py
L=gpurun_out/team_stream.log
: > $L
for e in ${ENVS:-"LDPC_TEAM_DYNAMIC=2" "LDPC_TEAM_DYNAMIC=1" "LDPC_TEAM_DYNAMIC=0" "LDPC_TEAM_CACHE_MIB=0"}; do
  echo "== $e" >> $L
  env ${e//,/ } BATCHES=${BATCHES:-256,512,1024,2048,4096} timeout -k 10 150 python $P >> $L 2>&1 || { echo "FAILED ($e)" >> $L; break; }
done
grep -v amdgpu.ids $L
