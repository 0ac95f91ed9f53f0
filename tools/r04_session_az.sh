#!/bin/bash
# hand-off copy asking only for the stragglers' part of every row (LDPC_DEFER_COPY_MASKED=1, new) against whole rows, alternating; parity of the new build first
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also'
S="500 \"python -m pytest tests -m gpu -x -q -k 'hand or defer or level or waterfall or realistic or c3 or ragged or full_batch'\""
for w in "c3_realistic" "c3_waterfall" "c3_realistic --llr"; do for v in mi355x v_copyall mi355x v_copyall; do S="$S 120 \"LDPC_MI355X_LIB=$C/libldpc_$v.so $B --workload $w\""; done; done
eval tools/gpu_session.sh r04az $S
tail -2 gpurun_out/r04az_1.log
k=1
for w in c3_realistic c3_waterfall c3_realistic_llr; do for v in masked whole masked whole; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04az_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-18s %-7s'%('$w','$v'), 'kernel %.2f step %.2f'%(r['kernel_ms'], d['ms_per_step']))"; done; done
