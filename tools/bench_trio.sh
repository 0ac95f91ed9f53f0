#!/bin/bash
# the three C3 workloads, N processes each, one line per run
N=${1:-2}
run() { echo -n "$WL $* :  "; env "$@" LDPC_PLACEMENT_VERBOSE=1 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f total_ms %.1f kernel_ms %.1f frac %.3f value %.0f' % (d['ms_per_step'], d['roofline']['pack_sweep_unpack_ms'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']))"; grep "workspace" /tmp/err.log | sed 's/^/      /'; }
for rep in $(seq $N); do
for WL in c3_full50 c3_waterfall c3_realistic; do run LDPC_X=0; done
done
