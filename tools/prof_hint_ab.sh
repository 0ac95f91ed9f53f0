#!/bin/bash
# does the private address region change what rocprofv3-profiled runs measure?  stats pass only, alternating
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
one() {  # tag, hint
  export LDPC_VMM_HINT_TIB=$2 LDPC_PLACEMENT_VERBOSE=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hint_$1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_hint_$1.log 2>&1
  echo "profiled  hint=$2: $(grep -o '"kernel_ms": [0-9.]*' $OUT/prof_hint_$1.log | head -1)  $(grep 'kept' $OUT/prof_hint_$1.log | sed 's/.*kept/kept/')"
  python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/plain_hint_$1.log 2>&1
  echo "plain     hint=$2: $(grep -o '"kernel_ms": [0-9.]*' $OUT/plain_hint_$1.log | head -1)  $(grep 'kept' $OUT/plain_hint_$1.log | sed 's/.*kept/kept/')"
}
one a 16; one b 0; one c 16; one d 0
