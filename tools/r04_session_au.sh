#!/bin/bash
# soak on the last binary: the suite once more and two more fuzz seeds
tools/gpu_session.sh r04au \
  900 'python -m pytest tests -m gpu -x -q' \
  480 'python tools/fuzz_parity.py 360 264575' \
  480 'python tools/fuzz_parity.py 360 300000'
tail -2 gpurun_out/r04au_1.log; tail -1 gpurun_out/r04au_2.log; tail -1 gpurun_out/r04au_3.log
