#!/bin/bash
# IRR with the two-halves update; the unpack_llr_kernel variants (tile -> XCD mapping, -log(v)) on the realistic LLR workload
C=$PWD/ldpcdecoders.jl_amd/csrc
tools/gpu_session.sh r04n \
  500 'python -m pytest tests -m gpu -x -q -k "irregular or expires or one_team_of_an_xcd or llr"' \
  400 'SIZES=16384,18432 python tools/irr_probe.py' \
  200 'BENCH_ARGS="--llr" SUFFIX=_llr_new tools/profile_workload.sh r04n c3_realistic' \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_unpack_old.so BENCH_ARGS='--llr' SUFFIX=_llr_old tools/profile_workload.sh r04n c3_realistic" \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_unpack_neglog.so BENCH_ARGS='--llr' SUFFIX=_llr_neglog tools/profile_workload.sh r04n c3_realistic" \
  200 'BENCH_ARGS="--llr" SUFFIX=_llr_new2 tools/profile_workload.sh r04n c3_realistic'
grep "^irregular" gpurun_out/r04n_2.log
for f in gpurun_out/prof_r04n_c3_realistic_llr_*/*/*_kernel_stats.csv; do echo $f; grep "unpack_llr\|bp_team_kernel" $f | cut -c1-70,180-300; done
grep -h '"metric"' gpurun_out/prof_r04n_*.log | cut -c1-200
