#!/bin/bash
# position chunks of one kind (all first edges in LDS / all in registers) on straight-line code: LDPC_TEAM_CHUNK_KINDS 1 (new) against 0, product-type builds, alternating
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
for w in "c3_full50" "c3_realistic" "reg410_16380" "reg36_16380" "c3_full50 --llr"; do for c in "NOTHING=1" "LDPC_MI355X_LIB=$C/libldpc_v_kinds_old.so" "NOTHING=1" "LDPC_MI355X_LIB=$C/libldpc_v_kinds_old.so"; do S="$S 120 \"$c $B --workload $w\""; done; done
eval tools/gpu_session.sh r04z $S
for k in $(seq 1 20); do grep -h '"metric"' gpurun_out/r04z_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print($k, 'new' if $k%2 else 'old', d['config']['workload'][:14], 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done
