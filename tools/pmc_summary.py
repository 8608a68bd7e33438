#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if "vrt::" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
