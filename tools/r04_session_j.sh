#!/bin/bash
# final form of the LLR capture (straight into the tile's rows): the suite, the bench line as the driver runs it, kernel stats of the
# LLR instantiation, a fuzz run
TAG=r04b
tools/gpu_session.sh r04j \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'python bench.py --steps 20 --warmup 5' \
  420 'python tools/fuzz_parity.py 300 31415'
for WL in c3_full50 c3_realistic c3_waterfall; do
  BENCH_ARGS="--llr" SUFFIX=_llr tools/profile_workload.sh $TAG $WL > gpurun_out/profile_${TAG}_${WL}_llr.log 2>&1
  grep -h '^{' gpurun_out/prof_${TAG}_${WL}_llr.log | tail -1 > gpurun_out/prof_${TAG}_${WL}_llr_bench.json
  tail -3 gpurun_out/profile_${TAG}_${WL}_llr.log | cut -c1-200
done
tail -3 gpurun_out/r04j_1.log; grep -h '^{' gpurun_out/r04j_2.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'value %.0f ms/step %.1f kernel_ms %.1f frac %.3f frac_of_bound %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_of_bound') or 0), r['phase_share_check_var_conv'], d['cpu_baseline']['gpu_matches_oracle_on_sample'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_faithful_1_thread']['value'])
    for k, v in d.get('also', {}).items(): print('   also', k, {a: (round(b, 4) if isinstance(b, float) and abs(b) > 1e-3 else b) for a, b in v.items() if a in ('ms_per_step', 'kernel_ms', 'value', 'frac', 'gpu_matches_oracle_on_sample', 'llr_max_abs_diff_vs_oracle', 'us_per_decode_median', 'mean_iters', 'osd_postprocessed_per_step', 'output_satisfies_syndrome_on_sample')})
"; tail -3 gpurun_out/r04j_3.log
