#!/bin/bash
# LDPC_TEAM_AHEAD (active lanes from which on a quiet tile's team runs ahead) 32 (default) against 16 and 8, alternating, experiments build
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
for w in c3_waterfall c3_realistic c3_full50 "c3_realistic --llr"; do for a in 32 16 32 16 8; do S="$S 120 \"LDPC_TEAM_AHEAD=$a $B --workload $w\""; done; done
eval tools/gpu_session.sh r04at $S
k=0
for w in c3_waterfall c3_realistic c3_full50 c3_realistic_llr; do for a in 32 16 32 16 8; do k=$((k+1)); printf "%-18s ahead %-3s " $w $a; grep -h '"metric"' gpurun_out/r04at_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('step %.2f kernel %.2f'%(d['ms_per_step'], r['kernel_ms']))"; done; done
