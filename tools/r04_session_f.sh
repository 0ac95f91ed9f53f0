#!/bin/bash
# round 4, sixth GPU call: the profiles of the round (rocprofv3 kernel stats + the two PMC passes for the headline, kernel stats for the
# other workloads and for the LLR-producing instantiation), then one fuzz run on the same binary
TAG=${1:-r04a}
tools/profile_c3.sh $TAG > gpurun_out/profile_${TAG}.log 2>&1
tail -6 gpurun_out/profile_${TAG}.log
for WL in c3_waterfall c3_realistic reg36_16380 wide_16000_10_5 reg39_16380 reg410_16380; do
  tools/profile_workload.sh $TAG $WL > gpurun_out/profile_${TAG}_${WL}.log 2>&1
  grep -h '^{' gpurun_out/prof_${TAG}_${WL}.log | tail -1 > gpurun_out/prof_${TAG}_${WL}_bench.json
  tail -3 gpurun_out/profile_${TAG}_${WL}.log | cut -c1-200
done
for WL in c3_full50 c3_realistic; do
  BENCH_ARGS="--llr" SUFFIX=_llr tools/profile_workload.sh $TAG $WL > gpurun_out/profile_${TAG}_${WL}_llr.log 2>&1
  grep -h '^{' gpurun_out/prof_${TAG}_${WL}_llr.log | tail -1 > gpurun_out/prof_${TAG}_${WL}_llr_bench.json
  tail -3 gpurun_out/profile_${TAG}_${WL}_llr.log | cut -c1-200
done
timeout -k 10 420 python tools/fuzz_parity.py 300 2718 > gpurun_out/fuzz_${TAG}.log 2>&1; echo "fuzz exit $?"; tail -4 gpurun_out/fuzz_${TAG}.log
