#!/bin/bash
# whole checks only for bit degree 3: the full-size tests of the regular pairs, then the two bench workloads (product build, default plan)
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
tools/gpu_session.sh r04w \
  600 'python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "regular or rows"' \
  200 "$B --workload reg36_16380" \
  200 "$B --workload reg39_16380"
tail -3 gpurun_out/r04w_1.log
grep -h '"metric"' gpurun_out/r04w_2.log gpurun_out/r04w_3.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:20], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('message_rows_on_chip_frac'))"
