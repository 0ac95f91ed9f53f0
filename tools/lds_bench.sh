#!/bin/bash
# the LDS-resident kernel's workloads (C2, the reference's test code, C5 BP+OSD), ms per step; with an argument: A/B
# against another build of the library
run() { for WL in c2_n1008 ref_1000_10_9 c5_bb72_bposd; do for B in 0 262144; do
  [ $WL = c5_bb72_bposd ] && [ $B != 0 ] && continue
  echo -n "$WL batch $B (0 = the workload's): "; python bench.py --workload $WL --batch $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.4f value %.3g' % (d['ms_per_step'], d['value']))"
done; done; }
if [ -n "$1" ]; then for rep in 1 2; do echo "== this build"; run; echo "== $1"; LDPC_MI355X_LIB=$PWD/$1 run; done; else run; fi
