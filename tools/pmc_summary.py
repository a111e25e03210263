#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: sum of each counter over dispatches / number of dispatches."""
import csv, sys, collections, glob
files = [f for a in sys.argv[1:] for f in glob.glob(a, recursive=True)]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    n = max(1, len(cnt[k]))
    print(k, "dispatches", n)
    for c, v in sorted(d.items()):
        print("   %-28s %.4g per dispatch" % (c, v / n))
