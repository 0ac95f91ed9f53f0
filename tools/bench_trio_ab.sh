#!/bin/bash
# the three C3 workloads under a list of environment settings (comma-separated assignments per run; default:
# as shipped against the tile kernel, LDPC_TEAM_CACHE_MIB=0 = teams only up to one tile per CU as in round 1), same box
run() { echo -n "$WL $1 :  "; env ${1//,/ } python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f total_ms %.1f kernel_ms %.1f frac %.3f value %.0f kernel %s' % (d['ms_per_step'], d['roofline']['pack_sweep_unpack_ms'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'], d['roofline']['kernel']))"; }
for WL in ${WLS:-c3_full50 c3_waterfall c3_realistic}; do
  for e in ${ENVS:-"X=0 LDPC_TEAM_CACHE_MIB=0"}; do run $e; done
done
