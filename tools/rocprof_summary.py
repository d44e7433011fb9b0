#!/usr/bin/env python3
"""Compact views of rocprofv3 output (replaces the former kstats.py / ktrace.py):
    python tools/rocprof_summary.py stats <*_kernel_stats.csv> [filter]     one line per kernel
    python tools/rocprof_summary.py trace <output dir> [filter]             per (kernel, grid): average / median duration
    python tools/rocprof_summary.py step  <output dir>                      the kernels of the last extraction step in launch
                                                                            order with the gaps between them"""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("orbgpu::", "").replace("void ", "")


def stats(path, flt=""):
    for r in csv.DictReader(open(path)):
        n = short(r["Name"])
        if flt and flt not in n:
            continue
        print("%-30s calls %5s avg %9.1f us  min %8.1f max %9.1f  tot %8.2f ms" % (
            n[:30], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            float(r["TotalDurationNs"]) / 1e6))


def rows_of(d):
    rows = []
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def trace(d, flt=""):
    acc = collections.defaultdict(list)
    for r in rows_of(d):
        n = short(r["Kernel_Name"])
        if flt and flt not in n:
            continue
        acc[(n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print("%-34s blocks %6d x %4d  n %4d  avg %8.1f us  median %8.1f us" % (k[0][:34], k[1], k[2], len(v), sum(v) / len(v),
                                                                                v[len(v) // 2]))


def step(d):
    rows = rows_of(d)
    names = [short(r["Kernel_Name"]) for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith("k_border0")]
    if len(starts) < 2:
        sys.exit("fewer than two extraction steps in the trace")
    i0, i1 = starts[-2], starts[-1]
    prev, gaps = None, 0.0
    for i in range(i0, i1):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        gaps += max(gap, 0.0)
        print("%-28s %8.1f us   gap %6.1f us" % (names[i][:28], (e - s) / 1e3, gap))
        prev = e
    print("step %.1f us, gaps %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3, gaps))


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "trace", "step"):
        sys.exit(__doc__)
    {"stats": stats, "trace": trace, "step": step}[sys.argv[1]](*sys.argv[2:])
