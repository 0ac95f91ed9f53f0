#!/usr/bin/env python3
"""Turn one tools/profile_c3.sh run (gpurun_out/prof_<tag>_{stats,fetch,write}*) into the summaries committed under
profiles/: <tag>_c3_full50_kernel_stats.csv (the --stats kernel table), <tag>_c3_full50_pmc_fetch_write.csv (the
counter rows of this library's kernels from the two PMC passes), <tag>_c3_full50_traffic.json (HBM bytes per launch of
the sweep kernel: 2 x FETCH_SIZE + WRITE_SIZE, units of 1 KiB, the gfx950 half-count of wide coalesced reads corrected
as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and <tag>_c3_full50_bench_under_rocprof.json.
Usage: tools/collect_profile.py <tag> [workload]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c3_full50"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
stem = os.path.join(P, f"{tag}_{workload}")

stats = glob.glob(os.path.join(G, f"prof_{tag}_stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, stem + "_kernel_stats.csv")
shutil.copy(os.path.join(G, f"prof_{tag}_bench.json"), stem + "_bench_under_rocprof.json")

b = json.load(open(stem + "_bench_under_rocprof.json"))
# the kernel the roofline is about: the first (fresh tiles) instantiation of the family bench.py names
family = b["roofline"].get("kernel", "bp_tile_kernel")
is_main = {"bp_tile_kernel": lambda k: "bp_tile_kernel" in k and "false, 512, false" in k,
           "bp_team_kernel": lambda k: "bp_team_kernel" in k and "false, 512, false" in k}.get(family, lambda k: family in k)

rows, header, total, names = [], None, {}, set()
for kind in ("fetch", "write"):
    f = glob.glob(os.path.join(G, f"prof_{tag}_{kind}", "*", "*_counter_collection.csv"))[0]
    with open(f, newline="") as fh:
        r = csv.reader(fh)
        h = next(r)
        header = header or h
        kn, cn, cv = h.index("Kernel_Name"), h.index("Counter_Name"), h.index("Counter_Value")
        for row in r:
            if "ldpc::" not in row[kn]:
                continue
            rows.append(row)
            if is_main(row[kn]):
                total.setdefault(row[cn], []).append(float(row[cv]))
                names.add(row[kn].split("(")[0])
with open(stem + "_pmc_fetch_write.csv", "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(header)
    w.writerows(rows)
fetch = sum(total["FETCH_SIZE"]) / len(total["FETCH_SIZE"])
write = sum(total["WRITE_SIZE"]) / len(total["WRITE_SIZE"])
traffic = (2.0 * fetch + write) * 1024.0
json.dump({
    "workload": workload,
    "kernel": family,
    "source": f"profiles/{tag}_{workload}_pmc_fetch_write.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
              f"{', '.join(sorted(names))}; the passes over the hand-off levels are empty at full-50; placement-probe rows left out)",
    "FETCH_SIZE_raw_kb": fetch,
    "WRITE_SIZE_raw_kb": write,
    "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); units of 1 KiB",
    "note": "the counters sit between the L2s and the fabric: bytes the Infinity Cache serves are counted like bytes from HBM",
    "traffic_bytes_per_launch": traffic,
}, open(stem + "_traffic.json", "w"), indent=1)
alg = b["roofline"]["alg_bytes_per_launch"]
print(f"{tag}: {family}: kernel_ms (bench, same run) {b['roofline']['kernel_ms']:.1f}, frac {b['roofline']['frac']:.4f}; "
      f"traffic {traffic / 1e12:.3f} TB = {traffic / alg:.4f} x algorithmic")
