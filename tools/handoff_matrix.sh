#!/bin/bash
# hand-off thresholds on the waterfall / realistic workloads (one process per line)
run() { echo -n "$* :  "; env "$@" python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f total_ms %.1f kernel_ms %.1f value %.0f' % (d['ms_per_step'], d['roofline']['pack_sweep_unpack_ms'], d['roofline']['kernel_ms'], d['value']))"; }
for WL in c3_waterfall c3_realistic; do
echo "== $WL"
run LDPC_X=0
run LDPC_DEFER_T1=0
run LDPC_DEFER_T0=24
run LDPC_DEFER_T0=32
run LDPC_DEFER_T0=32 LDPC_DEFER_T1=24
run LDPC_DEFER_T0=40 LDPC_DEFER_T1=32
run LDPC_DEFER_T0=48 LDPC_DEFER_T1=32
run LDPC_DEFER_T0=32 LDPC_NODE_TAKE_MAX=8192
run LDPC_DEFER_T0=32 LDPC_NODE_TAKE_MAX=256
done
WL=c3_full50
run LDPC_SLOT_MULT=1
run LDPC_SLOT_MULT=37
