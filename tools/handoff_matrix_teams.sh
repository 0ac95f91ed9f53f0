#!/bin/bash
# hand-off thresholds on the waterfall / realistic workloads with persistent teams (one process per line)
run() { echo -n "$* :  "; env "$@" python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f value %.0f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']))"; }
for WL in c3_waterfall c3_realistic; do
echo "== $WL"
run LDPC_X=0
run LDPC_DEFER_T0=8
run LDPC_DEFER_T0=12
run LDPC_DEFER_T0=24
run LDPC_DEFER_T0=32
run LDPC_DEFER_T0=24 LDPC_DEFER_T1=24
run LDPC_DEFER_T0=32 LDPC_DEFER_T1=24
run LDPC_DEFER_T1=8
run LDPC_DEFER_T1=0
run LDPC_NODE_TAKE_MAX=8192
run LDPC_NODE_TAKE_MAX=512
done
