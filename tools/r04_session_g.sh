#!/bin/bash
# where do the LLRs cost time at ~3 iterations a tile?  (timing probes: LDPC_TEAM_LLR_RAW 6 = the LLR instantiation alone, 7 = + capture,
# 8 = + copy-out only, 4 = both)
B="python bench.py --no-also --no-cpu-baseline --steps 4 --warmup 1 --workload c3_realistic"
tools/gpu_session.sh r04g \
  200 "$B" \
  200 "LDPC_TEAM_LLR_RAW=6 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=7 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=8 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=4 $B --llr" \
  200 "$B" \
  200 "LDPC_TEAM_LLR_RAW=6 $B --llr" \
  200 "LDPC_TEAM_LLR_RAW=4 LDPC_TEAM_AHEAD=0 $B --llr" \
  200 "LDPC_TEAM_AHEAD=0 $B"
for k in 1 2 3 4 5 6 7 8 9; do echo "== step $k"; grep -h '^{' gpurun_out/r04g_$k.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:14], 'ms/step %.1f kernel_ms %.2f total_ms %.1f' % (d['ms_per_step'], r['kernel_ms'], r['pack_sweep_unpack_ms']), r['phase_share_check_var_conv'])
"; done
