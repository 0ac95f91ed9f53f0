#!/bin/bash
# round 4, second GPU call: the new tests; the 32-bit LLR capture (TeamParams::llr_raw = 3) against the 64-bit one;
# the short division (-DLDPC_FAST_DIV=1 variant build) for parity and for time
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
FD=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_fastdiv.so
tools/gpu_session.sh r04b \
  600 'python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "short_division or eight_way"' \
  900 'python -m pytest tests/test_gpu_full_size.py -q -x -k "other_regular"' \
  200 "LDPC_TEAM_LLR_RAW=3 python bench.py --no-also --steps 3 --warmup 1 --llr" \
  200 "LDPC_TEAM_LLR_RAW=3 $B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_RAW=3 LDPC_TEAM_LLR_FOOTPRINT=0 $B --llr" \
  600 "LDPC_MI355X_LIB=$FD LDPC_MI355X_EXP_LIB=$FD python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -x" \
  200 "LDPC_MI355X_LIB=$FD LDPC_MI355X_EXP_LIB=$FD python bench.py --no-also --steps 4 --warmup 1" \
  200 "$B --steps 4" \
  200 "LDPC_MI355X_LIB=$FD LDPC_MI355X_EXP_LIB=$FD $B --steps 4" \
  200 "$B --steps 4" \
  200 "LDPC_MI355X_LIB=$FD LDPC_MI355X_EXP_LIB=$FD $B --workload c3_waterfall" \
  200 "$B --workload c3_waterfall" \
  200 "LDPC_MI355X_LIB=$FD LDPC_MI355X_EXP_LIB=$FD $B --workload c3_realistic" \
  200 "$B --workload c3_realistic"
for k in 3 4 5 7 8 9 10 11 12 13 14; do echo "== step $k"; grep -h '^{' gpurun_out/r04b_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.1f total_ms %.1f frac %.3f mean_iters %.2f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms'], r['frac'], d['config']['mean_iters']), r['phase_share_check_var_conv'], d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'), d.get('cpu_baseline', {}).get('llr_max_abs_diff_vs_oracle'))
"; done
