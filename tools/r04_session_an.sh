#!/bin/bash
# row stores write-through and agent-coherent (sc1; -DLDPC_STM_SC1=1, an experiments-build variant) against the experiments build: wide teams (n = 65536,
# 32768) and the headline, one box
C=$PWD/ldpcdecoders.jl_amd/csrc
V="LDPC_MI355X_LIB=$C/libldpc_v_sc1.so LDPC_MI355X_EXP_LIB=$C/libldpc_v_sc1.so"
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also --workload c3_full50'
tools/gpu_session.sh r04an \
  200 "N=65536 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "$V N=65536 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "N=32768 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  200 "$V N=32768 MODE=auto_vs_off python tools/wide_teams_probe.py" \
  120 "LDPC_TEAM_PRE=2 $B" \
  120 "$V LDPC_TEAM_PRE=2 $B" \
  120 "LDPC_TEAM_PRE=2 $B" \
  120 "$V LDPC_TEAM_PRE=2 $B"
for k in 1 2 3 4; do grep "^n " gpurun_out/r04an_$k.log | cut -c1-200; done
for k in 5 6 7 8; do grep -h '"metric"' gpurun_out/r04an_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print($k, 'kernel %.2f'%r['kernel_ms'], r.get('phase_share_check_var_conv'))"; done
