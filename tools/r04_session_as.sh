#!/bin/bash
# hand-off thresholds once more on the last kernels (experiments build): LDPC_DEFER_T0 / T1 at per 0.02 and 0.06
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
CFG=("LDPC_TEAM_PRE=2" "LDPC_DEFER_T0=8" "LDPC_DEFER_T0=12" "LDPC_DEFER_T0=24" "LDPC_DEFER_T0=32" "LDPC_DEFER_T0=16 LDPC_DEFER_T1=4" "LDPC_DEFER_T0=16 LDPC_DEFER_T1=16" "LDPC_TEAM_AHEAD=16" "LDPC_TEAM_AHEAD=48" "LDPC_TEAM_AHEAD_FROM=2")
S=""
for w in c3_realistic c3_waterfall; do for c in "${CFG[@]}"; do S="$S 120 \"$c $B --workload $w\""; done; done
eval tools/gpu_session.sh r04as $S
k=0
for w in c3_realistic c3_waterfall; do for c in "${CFG[@]}"; do k=$((k+1)); printf "%-14s %-36s " $w "$c"; grep -h '"metric"' gpurun_out/r04as_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('step %.2f kernel %.2f'%(d['ms_per_step'], r['kernel_ms']))"; done; done
