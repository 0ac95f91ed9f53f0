#!/bin/bash
# Which ingredient makes a profiled run crash inside exit() (SIGSEGV after rocprofv3 has written its files)?
# Runs the medium-batch bench (team kernel) under rocprofv3 --kernel-trace --stats three ways and greps the logs:
#   A  default build (plain team launches), decoder closed explicitly before exit
#   B  LDPC_TEAM_COOP_LAUNCH=1   (hipLaunchCooperativeKernel, as in round 1)
#   C  LDPC_TEAM_MAX=1           (no team kernel at all; the host-mapped fault word is still allocated)
# Usage (GPU box, repo root): tools/exit_crash_probe.sh   -> gpurun_out/exit_probe_{A,B,C}.log + summary on stdout
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # tag, env assignment
    local tag=$1; shift
    ( export "$@" LDPC_DUMMY=1; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exit_probe_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --batch 2048 --no-cpu-baseline > $OUT/exit_probe_$tag.log 2>&1; echo "exit code $?" >> $OUT/exit_probe_$tag.log )
    echo "== $tag ($*): $(grep -c SIGSEGV $OUT/exit_probe_$tag.log) SIGSEGV lines, $(tail -1 $OUT/exit_probe_$tag.log), kernel: $(grep -o '"kernel": "[a-z_]*"' $OUT/exit_probe_$tag.log | head -1)"
}
run A LDPC_X=0
run B LDPC_TEAM_COOP_LAUNCH=1
run C LDPC_TEAM_MAX=1
exit 0
