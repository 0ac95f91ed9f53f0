#!/bin/bash
# four checks / bits per lane in the pack / unpack kernels; the llr_cut pin test with the denormal band
tools/gpu_session.sh r04q \
  300 'python -m pytest tests -m gpu -x -q -k "alignment or cut_odds or ragged"' \
  900 'python -m pytest tests -m gpu -x -q' \
  200 'SUFFIX=_v4 tools/profile_workload.sh r04q c3_realistic' \
  200 'BENCH_ARGS="--llr" SUFFIX=_llr_v4 tools/profile_workload.sh r04q c3_realistic'
tail -3 gpurun_out/r04q_1.log; tail -3 gpurun_out/r04q_2.log
for f in gpurun_out/prof_r04q_c3_realistic_*/*/*_kernel_stats.csv; do echo $f; grep "unpack\|pack_syn\|bp_team_kernel\|bp_node" $f | awk -F'",' '{print substr($1,1,70), $2}'; done
grep -h '"metric"' gpurun_out/prof_r04q_*.log | cut -c1-250
