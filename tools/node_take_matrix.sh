#!/bin/bash
run() { echo -n "$WL $* :  "; env "$@" python bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for WL in c3_waterfall c3_realistic; do for t in 128 256 512 1024 2048; do run LDPC_NODE_TAKE_MAX=$t; done; done
WL=c3_realistic; run LDPC_PER=0; 
for p in 0.03 0.04 0.05; do echo -n "per $p: "; for t in 512 2048; do LDPC_NODE_TAKE_MAX=$t python bench.py --workload c3_realistic --per $p --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('take $t: %.1f ms' % d['ms_per_step'], end='   ')"; done; echo; done
