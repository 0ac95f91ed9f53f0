#!/bin/bash
# sc1 row stores in every team kernel: small and medium batches of the C3 code (teams over all XCDs), the suite, the default bench line
tools/gpu_session.sh r04ao \
  200 'CASES=16384:0.02,16384:0.10 AUTO=1 BATCHES=1,64,256,4096 python tools/smallbatch_probe.py' \
  900 'python -m pytest tests -m gpu -x -q' \
  300 'python bench.py'
grep "^n " gpurun_out/r04ao_1.log | cut -c1-200; tail -3 gpurun_out/r04ao_2.log; grep -h '"metric"' gpurun_out/r04ao_3.log | cut -c1-300
