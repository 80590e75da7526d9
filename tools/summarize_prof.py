#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    """kernel names without the namespaces / argument lists, so that the template arguments stay visible"""
    n = name.replace("void ", "").replace("mimc3::", "").replace("(anonymous namespace)::", "")
    n = n.replace("match_ncc_dlc_px<PxCfg<", "px<").replace("> >(MatchU8Args)", ">").replace("(MatchArgs)", "")
    i = n.find("(")
    return (n[:i] if i > 0 and not n.startswith("px<") else n).replace(", ", ",")


print(f"# rocprofv3 summary for {os.path.basename(out)}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel stats (--kernel-trace --stats):", os.path.relpath(f, out))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:40]:
        print("  {0:44.44s} calls={Calls:>5s} total_ns={TotalDurationNs:>12s} avg_ns={AverageNs:>14s} pct={Percentage}".format(short(r["Name"]), **r))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(r)
    # the matcher kernels by launch size: bench.py's io leg launches the headline kernel on CHUNKS of the grid too, so the
    # per-name average above mixes sizes -- the full-size launches are the ones `roofline.kernel_ms` times
    print("\n## matcher launches by grid size (kernel trace)")
    for k, v in by.items():
        if "match_ncc_dlc" not in k:
            continue
        sizes = defaultdict(list)
        for r in v:
            sizes[int(r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for g in sorted(sizes, reverse=True):
            d = sizes[g]
            print("  %-44.44s grid=%-9d calls=%4d avg_ns=%12.1f min_ns=%10d max_ns=%10d" % (short(k), g, len(d), sum(d) / len(d), min(d), max(d)))
    print("\n## dispatch resources (first dispatch of each kernel)")
    for k, v in by.items():
        r = v[0]
        keys = [c for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if c in r]
        print("  %-44.44s " % short(k) + " ".join(f"{c}={r[c]}" for c in keys))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        acc = defaultdict(lambda: defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"\n## PMC {os.path.basename(d)} (mean per dispatch)")
        for k, cs in acc.items():
            for c, vals in cs.items():
                print("  %-44.44s %-24s n=%d mean=%.6g max=%.6g" % (short(k), c, len(vals), sum(vals) / len(vals), max(vals)))
