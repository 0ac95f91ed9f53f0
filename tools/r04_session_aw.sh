#!/bin/bash
# the first iteration of a fresh tile without its check sweep (LDPC_TEAM_FUSE_FIRST=1, new) against with it: parity first, then alternating A/B
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S="700 \"python -m pytest tests -m gpu -x -q -k 'team or rows or c3 or regular or full_batch or waterfall or llr or wide or single'\""
W=("c3_realistic" "c3_waterfall" "c3_full50" "c3_realistic --llr" "reg36_16380" "reg410_16380")
for w in "${W[@]}"; do for v in mi355x v_nofuse mi355x v_nofuse; do S="$S 120 \"LDPC_MI355X_LIB=$C/libldpc_$v.so $B --workload $w\""; done; done
eval tools/gpu_session.sh r04aw $S
tail -3 gpurun_out/r04aw_1.log
k=1
for w in "${W[@]}"; do for v in fused plain fused plain; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04aw_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-22s'%'$w', '$v', 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done; done
