#!/bin/bash
# the team barrier on its arrival counter inside one XCD (LDPC_TEAM_BARRIER_COUNTER): A/B on one box, alternating
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also'
tools/gpu_session.sh r04t \
  200 "$B --workload c3_full50" \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_barrier_old.so $B --workload c3_full50" \
  200 "$B --workload c3_full50" \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_barrier_old.so $B --workload c3_full50" \
  200 "$B --workload reg36_16380" \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_barrier_old.so $B --workload reg36_16380" \
  200 "$B --workload c3_waterfall" \
  200 "LDPC_MI355X_LIB=$C/libldpc_v_barrier_old.so $B --workload c3_waterfall"
python - <<'PY'
import json,glob
for k in range(1,9):
    for l in open(f'gpurun_out/r04t_{k}.log'):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d['roofline']
            print(k, 'new' if k%2 else 'old', d['config']['workload'][:14], 'ms/step %.2f kernel %.2f'%(d['ms_per_step'], r['kernel_ms']), r.get('phase_share_check_var_conv'))
PY
