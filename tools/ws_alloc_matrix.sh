#!/bin/bash
# real-kernel A/B: how the workspace is backed (C3 full-50, 2 timed steps each, one process per line)
for rep in 1 2; do for a in malloc vmm:1024 vmm:1024:shuffle vmm:256:shuffle vmm:64 vmm:64:shuffle vmm:32768; do
  echo -n "LDPC_WS_ALLOC=$a  "
  LDPC_WS_ALLOC=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
