#!/bin/bash
# do the in-kernel phase stamps cost anything?  default build vs -DLDPC_PHASE_TICKS=0 (make variant), C3 full-50
run() { echo -n "$* :  "; env "$@" LDPC_PLACEMENT_VERBOSE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('kernel_ms %.1f frac %.3f' % (r['kernel_ms'], r['frac']), end='  ')"; grep "kept" /tmp/err.log | sed 's/.*kept/kept/'; }
for rep in 1 2 3; do
run LDPC_X=0
run LDPC_MI355X_LIB=$PWD/ldpcdecoders.jl_amd/csrc/libldpc_noticks.so
done
