#!/bin/bash
# round 4, third GPU call: LLRs through the teams' scratch rows (default: upper 32 bits of the odds; llr_exact: all 64),
# the t-form variant (division of :147 made by the variable sweep), then the whole suite
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
TF=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_tform.so
tools/gpu_session.sh r04c \
  900 'python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_bposd.py tests/test_golden.py tests/test_gpu_multi.py -q -x -k "llr or LLR or bposd or golden or bit_identical or eight_way or with_llrs"' \
  200 "python bench.py --no-also --steps 3 --warmup 1 --llr" \
  200 "$B --llr --llr-exact" \
  200 "$B --workload c3_realistic --llr" \
  200 "$B --workload c3_realistic --llr --llr-exact" \
  200 "$B --workload c3_waterfall --llr" \
  200 "$B --steps 4" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --steps 4" \
  200 "$B --steps 4" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --steps 4" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --workload c3_waterfall" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --workload c3_realistic" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --workload reg36_16380" \
  200 "$B --workload reg36_16380" \
  200 "LDPC_MI355X_LIB=$TF LDPC_MI355X_EXP_LIB=$TF $B --workload wide_16000_10_5" \
  200 "$B --workload wide_16000_10_5" \
  900 'python -m pytest tests -m gpu -x -q'
for k in 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do echo "== step $k"; grep -h '^{' gpurun_out/r04c_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.1f total_ms %.1f frac %.3f mean_iters %.2f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms'], r['frac'], d['config']['mean_iters']), r['phase_share_check_var_conv'], d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'), d.get('cpu_baseline', {}).get('llr_max_abs_diff_vs_oracle'))
"; done
tail -3 gpurun_out/r04c_1.log; tail -3 gpurun_out/r04c_17.log
