#!/bin/bash
# hand-off thresholds on the waterfall workload, final kernels (one process per line)
run() { echo -n "$* :  "; env "$@" LDPC_PLACEMENT_VERBOSE=1 python bench.py --workload c3_waterfall --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), end='  ')"; grep "kept" /tmp/err.log | sed 's/.*kept/kept/'; }
run LDPC_X=0
run LDPC_DEFER_T0=8
run LDPC_DEFER_T0=12
run LDPC_DEFER_T0=20
run LDPC_DEFER_T0=24
run LDPC_DEFER_T1=8
run LDPC_DEFER_T1=24
run LDPC_DEFER_T0=12 LDPC_DEFER_T1=12
run LDPC_NODE_TAKE_MAX=512
run LDPC_NODE_TAKE_MAX=8192
run LDPC_X=0
