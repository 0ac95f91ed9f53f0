#!/bin/bash
# product build against experiments build (same kernels but for two early-exit blocks of the fault injection), alternating on one box
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
CFG=("LDPC_TEAM_STATIC=3" "NOTHING=1" "LDPC_TEAM_STATIC=3" "NOTHING=1" "LDPC_TEAM_STATIC=3" "NOTHING=1")
S=""
for w in "c3_full50" "c3_full50 --llr" "c3_realistic" "reg410_16380"; do for c in "${CFG[@]:0:4}"; do S="$S 120 \"$c $B --workload $w\""; done; done
eval tools/gpu_session.sh r04y $S
for k in $(seq 1 16); do grep -h '"metric"' gpurun_out/r04y_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print($k, 'exp ' if $k%2 else 'prod', d['config']['workload'][:14], 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']), r.get('phase_share_check_var_conv'))"; done
