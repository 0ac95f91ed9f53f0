#!/bin/bash
# How busy are the vector ALUs under the C3 full-50 launch?  (each pass its own run, --pmc with --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
L=$OUT/valu_pmc.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_all.txt 2>/dev/null
: > $L
pass() {
  local tag=$1; shift
  rm -rf $OUT/pmc_$tag
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/pmc_$tag.log 2>&1
  echo "== $tag ($*)" >> $L
  python3 - $OUT/pmc_$tag >> $L <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "bp_team_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(acc.items()): print(f"   {k:32s} {v:.5g}")
PY
}
pass v1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass v2 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
pass v3 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass v4 SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
pass v5 SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
cat $L
