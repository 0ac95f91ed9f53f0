#!/bin/bash
# walking orders and static shares once more, with the pre-work in (experiments build, one box)
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also --workload c3_full50'
CFG=("LDPC_TEAM_FLIP=3" "LDPC_TEAM_FLIP=1" "LDPC_TEAM_FLIP=2" "LDPC_TEAM_FLIP=0" "LDPC_TEAM_FLIP=3" "LDPC_TEAM_FLIP=1" "LDPC_TEAM_STATIC=2" "LDPC_TEAM_STATIC=3" "LDPC_TEAM_REGS=28" "LDPC_TEAM_FLIP=3 LDPC_TEAM_PRE=2" "LDPC_TEAM_FLIP=1 LDPC_TEAM_PRE=3")
S=""
for c in "${CFG[@]}"; do S="$S 120 \"$c $B\""; done
eval tools/gpu_session.sh r04ak $S
k=0
for c in "${CFG[@]}"; do k=$((k+1)); printf "%-40s " "$c"; grep -h '"metric"' gpurun_out/r04ak_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('kernel %.2f'%r['kernel_ms'], r.get('phase_share_check_var_conv'))"; done
