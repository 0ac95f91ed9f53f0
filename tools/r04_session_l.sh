#!/bin/bash
tools/gpu_session.sh r04l \
  600 'SIZES=18432,20480,24576 python tools/irr_probe.py' \
  900 'python -m pytest tests -m gpu -x -q'
cat gpurun_out/r04l_1.log | grep "^irregular"; tail -4 gpurun_out/r04l_2.log
