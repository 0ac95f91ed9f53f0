#!/bin/bash
# LLR copy-out as a plain wide copy + unpack through the position map: parity (LLR tests), timing at per 0.02 / 0.06 / 0.10, then the suite
B="python bench.py --no-also --no-cpu-baseline --steps 4 --warmup 1"
tools/gpu_session.sh r04h \
  600 'python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_multi.py tests/test_golden.py -q -x -k "llr or with_llrs or eight_way or golden or auto_dispatch or bit_identical"' \
  200 "$B --workload c3_realistic" \
  200 "$B --workload c3_realistic --llr" \
  200 "LDPC_TEAM_LLR_RAW=8 $B --workload c3_realistic --llr" \
  200 "$B --workload c3_realistic --llr --llr-exact" \
  200 "$B --workload c3_waterfall" \
  200 "$B --workload c3_waterfall --llr" \
  200 "$B" \
  200 "$B --llr" \
  900 'python -m pytest tests -m gpu -x -q'
for k in 2 3 4 5 6 7 8 9; do echo "== step $k"; grep -h '^{' gpurun_out/r04h_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.2f total_ms %.1f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms']), r['phase_share_check_var_conv'])
"; done
tail -3 gpurun_out/r04h_1.log; tail -3 gpurun_out/r04h_10.log
