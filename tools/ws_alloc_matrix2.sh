#!/bin/bash
run() { echo -n "$* :  "; env "${@:2}" python bench.py --steps 2 --warmup 1 --no-cpu-baseline $1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
run "--defer-threshold=0" LDPC_X=0
run "--defer-threshold=-1" LDPC_X=0
run "--defer-threshold=0" LDPC_WS_ALLOC=vmm:1024
run "--defer-threshold=-1" LDPC_WS_ALLOC=vmm:1024
run "--defer-threshold=-1" LDPC_WS_ALLOC=malloc
run "--defer-threshold=0" LDPC_DEFER_T1=0
done
