#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes written by profiles/pmc_run.sh: per kernel, mean of every counter
over the dispatches of the non-counting build.  Usage: profiles/pmc_summary.py <dir> [> summary.txt]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}   n={len(v)}")
