#!/bin/bash
# check degree 10: the fused first iteration with / without the layout hints, against neither (three builds, alternating)
C=$PWD/ldpcdecoders.jl_amd/csrc
B='python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also'
S=""
for w in reg410_16380 wide_16000_10_5; do for v in v_nofuse mi355x v_fh10 v_nofuse mi355x v_fh10; do S="$S 120 \"LDPC_MI355X_LIB=$C/libldpc_$v.so $B --workload $w\""; done; done
eval tools/gpu_session.sh r04ax $S
k=0
for w in reg410 wide_10_5; do for v in plain fused fused+hints plain fused fused+hints; do k=$((k+1)); grep -h '"metric"' gpurun_out/r04ax_$k.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('%-12s %-12s'%('$w','$v'), 'kernel %.2f'%(r['kernel_ms']), r.get('phase_share_check_var_conv'))"; done; done
