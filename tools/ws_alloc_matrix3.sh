#!/bin/bash
# backing of the workspace without any search (LDPC_PLACEMENT_ROUNDS=1), C3 full-50, one process per line
for rep in 1 2 3; do for a in vmm:1024 vmm:64:shuffle vmm:256:shuffle vmm:1024:shuffle; do
  echo -n "LDPC_WS_ALLOC=$a  "
  LDPC_PLACEMENT_ROUNDS=1 LDPC_WS_ALLOC=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('kernel_ms %.1f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
