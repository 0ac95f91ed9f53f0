#!/bin/bash
# (4,8)-regular codes between the C3 size and the tile kernel's domain, 16,384 syndromes x 50 iterations: the planned
# teams (fully cached slots on 6-8 XCDs, or 8 partly cached ones) against eight teams forced, and against the tile kernel
L=gpurun_out/midsize_plan.log
: > $L
for n in 20480 24576 28672 40960 49152; do
  for e in "LDPC_TEAM_DEBUG=1" "LDPC_TEAM_XCDS=8,LDPC_TEAM_CACHE_MIB=2000" "LDPC_TEAM_CACHE_MIB=600"; do
    echo "== n $n $e" >> $L
    env ${e//,/ } N=$n LDPC_TEAM_DEBUG=1 BATCHES=16384 timeout -k 10 200 python tools/team_mall_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "^batch" | cut -c1-200 >> $L
  done
  echo "== n $n tile kernel" >> $L
  N=$n VARIANT=1 BATCHES=16384 timeout -k 10 200 python tools/team_mall_probe.py 2>&1 | grep -E "^batch" | cut -c1-120 >> $L
done
cat $L
