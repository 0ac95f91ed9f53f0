#!/bin/bash
# round-3 profiles in one go (run from the repo root on the GPU box): C3 full-50 with counters, then kernel stats of the
# waterfall / realistic workloads and of the two other regular codes that keep rows on chip
TAG=${1:-r03a}
tools/profile_c3.sh $TAG > gpurun_out/profile_${TAG}.log 2>&1
tail -6 gpurun_out/profile_${TAG}.log
for WL in c3_waterfall c3_realistic reg36_16380 wide_16000_10_5; do
  tools/profile_workload.sh $TAG $WL > gpurun_out/profile_${TAG}_${WL}.log 2>&1
  grep -h '^{' gpurun_out/prof_${TAG}_${WL}.log | tail -1 > gpurun_out/prof_${TAG}_${WL}_bench.json
  tail -3 gpurun_out/profile_${TAG}_${WL}.log | cut -c1-200
done
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "mall|hbm|dram|EA_RD|EA_WR|MALL" | head -40 ) > gpurun_out/counters_${TAG}.txt 2>&1
