#!/bin/bash
tools/gpu_session.sh r04m \
  700 'SIZES=16384,18432,20480,24576,32768 python tools/irr_probe.py' \
  900 'python -m pytest tests -m gpu -x -q'
cat gpurun_out/r04m_1.log | grep "^irregular"; tail -4 gpurun_out/r04m_2.log
