#!/bin/bash
# strays on chip or whole checks only, the other regular pairs (same box, alternating)
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
for w in reg39_16380 reg410_16380 wide_16000_10_5 reg36_16380; do for c in 1 2 1 2; do S="$S 200 \"LDPC_TEAM_CONCENTRATE=$c $B --workload $w\""; done; done
eval tools/gpu_session.sh r04v $S
python - <<'PY'
import json,glob
for k in range(1,17):
    for l in open(f'gpurun_out/r04v_{k}.log'):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d['roofline']
            print(k, 'strays' if k%2 else 'whole ', d['config']['workload'][:18], 'ms/step %.2f kernel %.2f'%(d['ms_per_step'], r['kernel_ms']), r.get('phase_share_check_var_conv'), '%.4f'%r.get('message_rows_on_chip_frac'))
PY
