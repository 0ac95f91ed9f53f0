#!/bin/bash
# knob sweep of the team kernel on the headline workload, one box (experiments build for the env knobs; variant builds for LDPC_TEAM_SLEEP)
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also --workload c3_full50'
CFG=("LDPC_TEAM_DEBUG=0" "LDPC_TEAM_STATIC=2" "LDPC_TEAM_STATIC=4" "LDPC_TEAM_STATIC=1" "LDPC_TEAM_FLIP=0" "LDPC_TEAM_FLIP=1" "LDPC_TEAM_FLIP=2" "LDPC_TEAM_DEBUG=0" \
     "LDPC_TEAM_PAIRS=0" "LDPC_TEAM_PAIRS=1" "LDPC_TEAM_PAIRS=2" "LDPC_TEAM_DYNAMIC=0" "LDPC_TEAM_REGS=24" "LDPC_TEAM_REGS=28" "LDPC_TEAM_AHEAD_FROM=2" "LDPC_TEAM_DEBUG=0" \
     "LDPC_MI355X_LIB=$C/libldpc_v_sleep4.so" "LDPC_MI355X_LIB=$C/libldpc_v_sleep1.so" "LDPC_MI355X_LIB=$C/libldpc_mi355x.so" "LDPC_MI355X_LIB=$C/libldpc_v_sleep4.so" "LDPC_MI355X_LIB=$C/libldpc_v_sleep1.so" "LDPC_MI355X_LIB=$C/libldpc_mi355x.so")
S=""
for c in "${CFG[@]}"; do S="$S 120 \"$c $B\""; done
eval tools/gpu_session.sh r04x $S
k=0
for c in "${CFG[@]}"; do k=$((k+1)); printf "%-55s " "${c##*/}"; grep -h '"metric"' gpurun_out/r04x_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('kernel %.2f'%r['kernel_ms'], r.get('phase_share_check_var_conv'))"; done
