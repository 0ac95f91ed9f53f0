#!/bin/bash
# strays on chip (rows of checks that are only partly on chip) or whole checks only: LDPC_TEAM_CONCENTRATE=2, same box, alternating (experiments build both times)
B='python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also'
tools/gpu_session.sh r04u \
  200 "LDPC_TEAM_CONCENTRATE=1 $B --workload c3_full50" \
  200 "LDPC_TEAM_CONCENTRATE=2 $B --workload c3_full50" \
  200 "LDPC_TEAM_CONCENTRATE=1 $B --workload c3_full50" \
  200 "LDPC_TEAM_CONCENTRATE=2 $B --workload c3_full50" \
  200 "LDPC_TEAM_CONCENTRATE=1 $B --workload reg36_16380" \
  200 "LDPC_TEAM_CONCENTRATE=2 $B --workload reg36_16380" \
  200 "LDPC_TEAM_CONCENTRATE=1 $B --workload c3_waterfall" \
  200 "LDPC_TEAM_CONCENTRATE=2 $B --workload c3_waterfall"
python - <<'PY'
import json,glob
for k in range(1,9):
    for l in open(f'gpurun_out/r04u_{k}.log'):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d['roofline']
            print(k, 'strays' if k%2 else 'whole ', d['config']['workload'][:14], 'ms/step %.2f kernel %.2f'%(d['ms_per_step'], r['kernel_ms']), r.get('phase_share_check_var_conv'), r.get('message_rows_on_chip_frac'))
PY
