#!/bin/bash
# rocprofv3 kernel stats of one bench workload (no counters).  Usage: [BENCH_ARGS="--llr"] [SUFFIX=_llr] tools/profile_workload.sh <tag> <workload>
TAG=${1:-run}; WL=${2:-c3_waterfall}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WLS=${WL}${SUFFIX}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_${WLS} -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-also $BENCH_ARGS > $OUT/prof_${TAG}_${WLS}.log 2>&1
head -12 $OUT/prof_${TAG}_${WLS}/*/*_kernel_stats.csv | cut -c1-60,200-330
