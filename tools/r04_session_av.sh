#!/bin/bash
# block placement by the ext-TSP model (-mllvm -enable-ext-tsp-block-placement, a variant of the product build) against the product build, alternating
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
W=("c3_full50" "c3_realistic" "reg36_16380" "reg410_16380" "wide_16000_10_5" "c3_full50 --llr")
for w in "${W[@]}"; do for v in mi355x v_tsp mi355x v_tsp; do S="$S 120 \"LDPC_MI355X_LIB=$C/libldpc_$v.so $B --workload $w\""; done; done
eval tools/gpu_session.sh r04av $S
k=0
for w in "${W[@]}"; do for v in prod tsp prod tsp; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04av_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-22s'%'$w', '$v', 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done; done
