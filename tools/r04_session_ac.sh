#!/bin/bash
# final binary of the round: the suite, the headline profile (kernel stats), the default bench line, the (5,10) / (4,10) workloads once more
tools/gpu_session.sh r04ac \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'tools/profile_workload.sh r04d c3_full50' \
  300 'python bench.py' \
  120 'python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also --workload wide_16000_10_5' \
  120 'python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also --workload reg410_16380'
tail -3 gpurun_out/r04ac_1.log
grep "bp_team_kernel" gpurun_out/prof_r04d_c3_full50/*/*_kernel_stats.csv | cut -c1-60,150-330
grep -h '"metric"' gpurun_out/r04ac_3.log | cut -c1-1500
for k in 4 5; do grep -h '"metric"' gpurun_out/r04ac_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:24], 'kernel %.2f'%r['kernel_ms'], r['frac'])"; done
