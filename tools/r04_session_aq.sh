#!/bin/bash
# two more fuzz runs on the last binary (sc1 row stores, barrier-shadow work, stray knobs in the draw)
tools/gpu_session.sh r04aq \
  520 'python tools/fuzz_parity.py 400 173205' \
  520 'python tools/fuzz_parity.py 400 223606'
tail -2 gpurun_out/r04aq_1.log; tail -2 gpurun_out/r04aq_2.log
