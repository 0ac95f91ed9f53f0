#!/bin/bash
# last binary of the round: the suite, a fuzz run (irregular mid-size graphs in the draw), the default bench line
tools/gpu_session.sh r04al \
  900 'python -m pytest tests -m gpu -x -q' \
  560 'python tools/fuzz_parity.py 420 141421' \
  300 'python bench.py'
tail -3 gpurun_out/r04al_1.log; tail -3 gpurun_out/r04al_2.log; grep -h '"metric"' gpurun_out/r04al_3.log | cut -c1-400
