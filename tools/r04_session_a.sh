#!/bin/bash
# round 4, first GPU call: the suite on the new binary (bounded waits, M0, raw LLR), the headline, and the LLR-producing
# instantiation at C3 size under the variants of TeamParams::llr_raw and of the plan's footprint rule
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
tools/gpu_session.sh r04a \
  700 'python -m pytest tests -m gpu -x -q' \
  240 'python bench.py --steps 5 --warmup 2 --no-also' \
  240 "python bench.py --no-also --steps 3 --warmup 1 --llr" \
  200 "LDPC_TEAM_LLR_RAW=0 LDPC_TEAM_LLR_FOOTPRINT=0 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=1 LDPC_TEAM_LLR_FOOTPRINT=0 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=2 LDPC_TEAM_LLR_FOOTPRINT=0 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=2 LDPC_TEAM_LLR_FOOTPRINT=1 $B --llr" \
  200 "$B --workload c3_realistic" \
  200 "$B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_RAW=0 LDPC_TEAM_LLR_FOOTPRINT=0 $B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_RAW=1 LDPC_TEAM_LLR_FOOTPRINT=0 $B --workload c3_realistic --llr" \
  420 'python tools/fuzz_parity.py 300 9595'
for k in 2 3 4 5 6 7 8 9 10 11; do echo "== step $k"; grep -h '^{' gpurun_out/r04a_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.1f total_ms %.1f frac %.3f mean_iters %.2f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms'], r['frac'], d['config']['mean_iters']), d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'), d.get('cpu_baseline', {}).get('llr_max_abs_diff_vs_oracle'))
"; done
