#!/bin/bash
# code-layout variants of the team kernel (same arithmetic): A product, B with the two early-exit blocks of the fault injection, C likely-hints on the hot
# chunk conditions, D = B + C, E = chunk kinds + hints; one box, two rounds
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
for w in "c3_full50" "reg36_16380" "reg410_16380" "c3_realistic" ; do for round in 1 2; do for v in mi355x v_B v_C v_D v_E; do S="$S 120 \"LDPC_MI355X_LIB=$C/libldpc_$v.so $B --workload $w\""; done; done; done
eval tools/gpu_session.sh r04aa $S
k=0
for w in c3_full50 reg36 reg410 c3_realistic; do for round in 1 2; do for v in A B C D E; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04aa_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('$w', '$v', 'kernel %.2f'%(r['kernel_ms']), r.get('phase_share_check_var_conv'))"; done; done; done
