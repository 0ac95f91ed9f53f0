#!/bin/bash
# after llr_cut / the unrolled unpack / IRR in two halves: the suite, the default bench line, a fuzz run with irregular mid-size graphs in the mix
tools/gpu_session.sh r04p \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'python bench.py' \
  560 'python tools/fuzz_parity.py 420 27182'
tail -3 gpurun_out/r04p_1.log; grep -h '"metric"' gpurun_out/r04p_2.log | cut -c1-3000; tail -3 gpurun_out/r04p_3.log
