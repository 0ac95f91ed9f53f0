#!/bin/bash
# last binary (fused first iteration, hints for every check degree): the suite, the headline profile with the PMC passes, the default bench line, a fuzz run
tools/gpu_session.sh r04ay \
  900 'python -m pytest tests -m gpu -x -q' \
  700 'tools/profile_c3.sh r04f' \
  300 'python bench.py' \
  480 'python tools/fuzz_parity.py 360 316227'
tail -2 gpurun_out/r04ay_1.log; tail -2 gpurun_out/r04ay_2.log; grep -h '"metric"' gpurun_out/r04ay_3.log | cut -c1-300; tail -1 gpurun_out/r04ay_4.log
