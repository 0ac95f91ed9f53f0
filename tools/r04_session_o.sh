#!/bin/bash
# llr_cut (LLRs without the library log) + the unrolled unpack_llr_kernel: the pin test, the suite, the LLR workloads
tools/gpu_session.sh r04o \
  300 'python -m pytest tests -m gpu -x -q -k "llr or irregular"' \
  200 'BENCH_ARGS="--llr" SUFFIX=_llr tools/profile_workload.sh r04o c3_realistic' \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'python bench.py --workload c3_full50 --llr --steps 3 --warmup 1 --no-cpu-baseline --no-also' \
  200 'python bench.py --workload c3_realistic --steps 5 --warmup 1 --no-cpu-baseline --no-also' \
  200 'python bench.py --workload c3_realistic --llr --steps 5 --warmup 1 --no-cpu-baseline --no-also'
for f in gpurun_out/prof_r04o_c3_realistic_llr/*/*_kernel_stats.csv; do grep "unpack_llr\|bp_team_kernel" $f | awk -F'",' '{print substr($1,1,60), $2}'; done
tail -3 gpurun_out/r04o_1.log; tail -3 gpurun_out/r04o_3.log
grep -h '"metric"' gpurun_out/r04o_4.log gpurun_out/r04o_5.log gpurun_out/r04o_6.log | cut -c1-400
