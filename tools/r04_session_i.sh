#!/bin/bash
# LLR capture straight into the tile's rows (wide stores, position layout) against the teams' scratch rows + copy-out
B="python bench.py --no-also --no-cpu-baseline --steps 4 --warmup 1"
tools/gpu_session.sh r04i \
  200 "$B --workload c3_realistic" \
  200 "LDPC_TEAM_LLR_SCRATCH=1 $B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 $B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 $B --workload c3_realistic --llr --llr-exact" \
  200 "$B --workload c3_waterfall" \
  200 "LDPC_TEAM_LLR_SCRATCH=1 $B --workload c3_waterfall --llr" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 $B --workload c3_waterfall --llr" \
  200 "$B" \
  200 "LDPC_TEAM_LLR_SCRATCH=1 $B --llr" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 $B --llr" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 $B --llr --llr-exact" \
  200 "LDPC_TEAM_LLR_SCRATCH=0 python bench.py --no-also --steps 2 --warmup 1 --llr --workload c3_realistic"
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do echo "== step $k"; grep -h '^{' gpurun_out/r04i_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.2f total_ms %.1f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms']), r['phase_share_check_var_conv'], d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'), d.get('cpu_baseline', {}).get('llr_max_abs_diff_vs_oracle'))
"; done
