#!/bin/bash
# on-chip checks updated between arriving at the barrier behind the variable sweep and waiting at it (TeamParams::pre): parity first, then
# LDPC_TEAM_PRE = 0 ... 4 on the headline workload and others (experiments build), alternating
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S="600 \"python -m pytest tests -m gpu -x -q -k 'team or rows or c3 or regular or full_batch or waterfall'\""
for w in "c3_full50" "c3_realistic" "reg36_16380" "reg410_16380"; do for v in 0 2 0 2 1 3 4; do S="$S 120 \"LDPC_TEAM_PRE=$v $B --workload $w\""; done; done
eval tools/gpu_session.sh r04aj $S
tail -3 gpurun_out/r04aj_1.log
k=1
for w in c3_full50 c3_realistic reg36 reg410; do for v in 0 2 0 2 1 3 4; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04aj_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-14s'%'$w', 'pre $v', 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done; done
