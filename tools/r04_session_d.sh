#!/bin/bash
# round 4, fourth GPU call: the suite on the binary with the t form and the wide LLR capture, the whole bench line with
# its new also{} entries, LLR A/B on one box, wide teams on n = 65536
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
tools/gpu_session.sh r04d \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'python bench.py --steps 5 --warmup 2' \
  200 "$B --steps 4" \
  200 "$B --llr" \
  200 "$B --llr --llr-exact" \
  200 "$B --workload c3_realistic" \
  200 "$B --workload c3_realistic --llr" \
  200 "$B --workload c3_waterfall" \
  200 "$B --workload c3_waterfall --llr" \
  600 "python tools/wide_teams_probe.py" \
  300 "N=32768 python tools/wide_teams_probe.py"
for k in 2 3 4 5 6 7 8 9; do echo "== step $k"; grep -h '^{' gpurun_out/r04d_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.1f total_ms %.1f frac %.3f mean_iters %.2f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms'], r['frac'], d['config']['mean_iters']), r['phase_share_check_var_conv'], d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'), d.get('cpu_baseline', {}).get('llr_max_abs_diff_vs_oracle'))
    for k, v in d.get('also', {}).items(): print('   also', k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ('ms_per_step', 'kernel_ms', 'value', 'frac', 'gpu_matches_oracle_on_sample', 'llr_max_abs_diff_vs_oracle', 'us_per_decode_median', 'mean_iters', 'osd_postprocessed_per_step', 'output_satisfies_syndrome_on_sample')})
"; done
tail -3 gpurun_out/r04d_1.log; cat gpurun_out/r04d_10.log gpurun_out/r04d_11.log | grep -v amdgpu.ids
