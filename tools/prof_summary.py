#!/usr/bin/env python3
"""Reduce a rocprofv3 output directory (csv) to a small text summary for profiles/.
usage: prof_summary.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, d))
    for row in csv.DictReader(open(f)):
        print("  %-90s calls=%s total_ns=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
            row.get("Name", "")[:90], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
            row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    by = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        # one line per (kernel, launch size): a kernel launched at several sizes has several average durations
        grid = row.get("Grid_Size") or row.get("Grid_Size_X")
        name = row.get("Kernel_Name", "") + ("  [grid %s]" % grid if grid else "")
        by[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[name] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), grid,
                      row.get("Workgroup_Size") or row.get("Workgroup_Size_X"))
    print("== kernel trace:", os.path.relpath(f, d))
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        if flt and flt not in name:
            continue
        v.sort()
        print("  %-118s n=%d avg_ns=%.0f med_ns=%d min_ns=%d max_ns=%d vgpr=%s sgpr=%s lds=%s grid=%s wg=%s" % (
            name[:60] + " .. " + name[-52:] if len(name) > 118 else name, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1], *meta[name]))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if flt and flt not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", os.path.relpath(f, d))
    for name, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("  %-70s %-28s n=%d avg=%.1f min=%.1f max=%.1f" % (name[:70], c, len(v), sum(v) / len(v), min(v), max(v)))
