#!/bin/bash
# real-kernel A/B: slot multiplier x placement search (C3 full-50, 2 timed steps each)
for m in 1 37 331; do for c in 1 12; do
  echo "== LDPC_SLOT_MULT=$m LDPC_PLACEMENT_CANDIDATES=$c"
  LDPC_SLOT_MULT=$m LDPC_PLACEMENT_CANDIDATES=$c python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
