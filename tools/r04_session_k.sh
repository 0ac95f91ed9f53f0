#!/bin/bash
B="python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1"
tools/gpu_session.sh r04k \
  300 'python -m pytest tests/test_gpu_reference_api.py -q -x -k "expires"' \
  600 'python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -x -k "irregular or auto_dispatch or wide_degree or llr_precision"' \
  300 'bash -c "time python bench.py --steps 20 --warmup 5 > gpurun_out/r04k_bench.json"' \
  600 'python tools/irr_probe.py' \
  420 'python tools/fuzz_parity.py 300 1618'
tail -5 gpurun_out/r04k_1.log; tail -5 gpurun_out/r04k_2.log; tail -5 gpurun_out/r04k_3.log; cat gpurun_out/r04k_4.log | grep -v amdgpu; tail -4 gpurun_out/r04k_5.log
