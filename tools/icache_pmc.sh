#!/bin/bash
# Does the team kernel with rows in registers (95 KB of code against 61 KB) miss the instruction cache?  SQC counters of
# the C3 full-50 launch, as shipped and with LDPC_TEAM_REGS=0 (each pass its own run, --pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
L=$OUT/icache_pmc.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_CACHE|WAIT_INST|BUSY_CY|WAVE_CYCLES|INSTS_VALU|INST_CYCLES)[A-Z_0-9]*)" | sort -u > $OUT/icache_counters.txt
: > $L
pass() {   # tag, counters...
  local tag=$1; shift
  rm -rf $OUT/pmc_$tag
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/pmc_$tag.log 2>&1
  echo "== $tag ($*) LDPC_TEAM_REGS=${LDPC_TEAM_REGS:-default}" >> $L
  python3 - $OUT/pmc_$tag >> $L <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "bp_team_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(acc.items()): print(f"   {k:28s} {v:.4g}")
PY
}
for regs in default 0; do
  if [ $regs = default ]; then unset LDPC_TEAM_REGS; else export LDPC_TEAM_REGS=$regs; fi
  pass ic_$regs SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
  pass sq_$regs SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU
done
cat $L
