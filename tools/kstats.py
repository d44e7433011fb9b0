#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: python tools/kstats.py <csv> [filter]"""
import csv
import sys

f = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("orbgpu::", "").replace("void ", "")
    if flt and flt not in n:
        continue
    print("%-30s calls %5s avg %9.1f us  min %8.1f max %9.1f  tot %8.2f ms" % (
        n[:30], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
        float(r["TotalDurationNs"]) / 1e6))
