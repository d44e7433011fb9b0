"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace output directory: python tools/ktrace.py DIR [filter]"""
import collections
import csv
import glob
import sys

d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("orbgpu::", "").split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in n:
            continue
        d[(n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-34s blocks %6d x %4d  n %4d  avg %8.1f us  median %8.1f us" % (k[0][:34], k[1], k[2], len(v), sum(v) / len(v), v[len(v) // 2]))
