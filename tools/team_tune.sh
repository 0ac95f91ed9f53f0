#!/bin/bash
# full-batch C3 (bench.py c3_full50) through the persistent teams under a list of environment settings
# (comma-separated assignments per run), one line per run
L=gpurun_out/team_tune.log
: > $L
for e in ${ENVS:-"X=0"}; do
  echo -n "$e :  " >> $L
  env ${e//,/ } timeout -k 10 200 python bench.py --workload ${WL:-c3_full50} --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.1f kernel_ms %.1f frac %.3f value %.0f kernel %s phases %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'], d['roofline']['kernel'], d['roofline']['phase_share_check_var_conv']))" >> $L 2>&1 || { echo "FAILED" >> $L; tail -3 /tmp/err.log >> $L; }
done
cat $L
