#!/bin/bash
# round 4, fifth GPU call: wide teams by themselves (the automatic rule) against the plan without them at sizes between the C3 code
# and n = 131072; LLRs once more (copy-out two chunks at a time); the suite
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
P="MODE=auto_vs_off python tools/wide_teams_probe.py"
tools/gpu_session.sh r04e \
  600 'python -m pytest tests/test_gpu_parity.py -q -x -k "auto_dispatch or llr_precision"' \
  200 "$B --llr" \
  200 "$B --workload c3_realistic --llr" \
  200 "$B --workload c3_realistic" \
  200 "$B --workload c3_waterfall --llr" \
  200 "N=20480 $P" \
  200 "N=24576 ALSO=8 $P" \
  200 "N=28672 ALSO=8 $P" \
  200 "N=40960 ALSO=2,4 $P" \
  300 "N=49152 ALSO=2 TILE=1 $P" \
  300 "N=98304 TILE=1 $P" \
  300 "N=131072 BATCH=8192 TILE=1 $P" \
  300 "N=65536 BATCH=1024 $P" \
  300 "N=32768 BATCH=2048 $P" \
  900 'python -m pytest tests -m gpu -x -q'
for k in 2 3 4 5; do echo "== step $k"; grep -h '^{' gpurun_out/r04e_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.1f total_ms %.1f frac %.3f mean_iters %.2f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms'], r['frac'], d['config']['mean_iters']), r['phase_share_check_var_conv'])
"; done
tail -3 gpurun_out/r04e_1.log; cat gpurun_out/r04e_{6,7,8,9,10,11,12,13,14}.log | grep "^n [0-9]"; tail -3 gpurun_out/r04e_15.log
