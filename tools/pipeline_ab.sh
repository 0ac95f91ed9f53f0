#!/bin/bash
# A/B of the software-pipelined variable sweep (C3 full-50); the probe time of the kept workspace group is printed
# with every run because the placement class moves the result as much as the change under test
run() { echo -n "$* :  "; env "$@" LDPC_PLACEMENT_VERBOSE=1 python bench.py --workload ${WL:-c3_full50} --steps 2 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('kernel_ms %.1f frac %.3f shares %s' % (r['kernel_ms'], r['frac'], r['phase_share_check_var_conv']), end='  ')"; grep "kept" /tmp/err.log | sed 's/.*kept/kept/'; }
for rep in 1 2 3; do
run LDPC_X=0
run LDPC_NO_VAR_PIPELINE=1
done
WL=c3_waterfall run LDPC_X=0
WL=c3_waterfall run LDPC_NO_VAR_PIPELINE=1
