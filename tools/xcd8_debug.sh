#!/bin/bash
# per-team member statistics (LDPC_TEAM_DEBUG) of the C3 code on 7 against 8 XCDs, rows in LDS, 256 tiles
L=gpurun_out/xcd8_debug.log
: > $L
for x in 7 8; do
  echo "== XCDS=$x" >> $L
  LDPC_TEAM_DEBUG=1 LDPC_TEAM_XCDS=$x LDPC_TEAM_CACHE_MIB=400 BATCHES=16384 timeout -k 10 120 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | tail -14 >> $L
done
cat $L
