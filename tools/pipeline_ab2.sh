#!/bin/bash
run() { echo -n "batch $B $* :  "; env "$@" python bench.py --workload c3_full50 --batch $B --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('%s kernel_ms %.1f frac %.3f shares %s' % (r['kernel'], r['kernel_ms'], r['frac'], r['phase_share_check_var_conv']))"; }
for B in 16384 24576 32768; do
run LDPC_X=0
run LDPC_NO_VAR_PIPELINE=1
done
