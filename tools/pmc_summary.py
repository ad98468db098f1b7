#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under a directory: per-kernel mean counter values + durations."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "snappy" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "snappy" in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(agg):
    if "compress" not in k:
        continue
    d = sorted(dur[k])
    print("==", k, "median ms", d[len(d) // 2] if d else None)
    for c, v in sorted(agg[k].items()):
        v = sorted(v)[-2:]
        print(f"   {c:26s} {sum(v) / len(v):18.0f}")
