#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print(",".join(row))
print()
print("== per-kernel trace: launches, avg duration (ns), resources ==")
for f in find("trace/**/*kernel_trace.csv"):
    agg = defaultdict(list)
    res = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            agg[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            res[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-60s n=%d avg_ns=%.0f total_ms=%.3f %s" % (name[:60], len(d), sum(d) / len(d), sum(d) / 1e6, res[name]))
print()
print("== PMC counters: per-kernel average over dispatches ==")
for d in sorted(os.path.basename(p) for p in glob.glob(os.path.join(out, "pmc_*"))):
    for f in find(d + "/**/*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row.get("Kernel_Name", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, cs in agg.items():
            if name.startswith("__amd_rocclr") or "transpose" in name or "init_spins" in name or "spins_in" in name or "spins_out" in name:
                continue
            print("[%s] %s" % (d, name[:70]))
            for c, v in sorted(cs.items()):
                print("    %-28s avg=%.6g  n=%d" % (c, sum(v) / len(v), len(v)))
