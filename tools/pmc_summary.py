#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (average per dispatch)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    if "wino" not in k and "conv3x3" not in k:
        continue
    print(k, "dispatch_avg_ns=%.0f" % (sum(dur[k]) / len(dur[k])))
    for c, v in sorted(agg[k].items()):
        print("   %-32s %.4g" % (c, sum(v) / len(v)))
