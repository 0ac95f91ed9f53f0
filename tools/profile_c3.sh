#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel stats + the two PMC passes
# for the bench workload.  Usage: tools/profile_c3.sh <tag>   -> gpurun_out/prof_<tag>_*
# Counters are collected in their own runs (--pmc with --kernel-trace only), as the pool requires.
set -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $OUT/prof_${TAG}_stats.log 2>&1
grep '^{' $OUT/prof_${TAG}_stats.log | tail -1 > $OUT/prof_${TAG}_bench.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/prof_${TAG}_write.log 2>&1
cat $OUT/prof_${TAG}_stats/*/*_kernel_stats.csv | head -4 | cut -c1-160
grep -h -E "bp_tile_kernel|bp_team_kernel" $OUT/prof_${TAG}_fetch/*/*_counter_collection.csv $OUT/prof_${TAG}_write/*/*_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
