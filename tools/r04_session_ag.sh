#!/bin/bash
# stray bits at the end of a member's positions (LDPC_TEAM_STRAYS_LAST=1, the new default) against dealt by number (0): experiments build both times, alternating
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
W=("c3_full50" "c3_realistic" "reg410_16380" "wide_16000_10_5" "c3_full50 --llr")
for w in "${W[@]}"; do for v in 1 0 1 0; do S="$S 120 \"LDPC_TEAM_STRAYS_LAST=$v $B --workload $w\""; done; done
eval tools/gpu_session.sh r04ag $S
k=0
for w in "${W[@]}"; do for v in last bynum last bynum; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04ag_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-22s'%'$w', '$v', 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done; done
