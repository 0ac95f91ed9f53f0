#!/bin/bash
# per-team sweep times of the C3 code by the number of XCDs that host a team (1 team each): is a team slowed by the others?
L=gpurun_out/xcd_scaling.log
: > $L
for x in 1 2 4 6 8; do
  echo "== LDPC_TEAM_XCDS=$x" >> $L
  LDPC_TEAM_XCDS=$x LDPC_TEAM_DEBUG=1 BATCHES=$((x * 1024)) timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "batch" >> $L
done
cat $L
