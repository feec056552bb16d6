#!/usr/bin/env python
"""Sum rocprofv3 PMC counters per kernel name: python tools/pmc_sum.py <dir> [substring]"""
import collections, csv, glob, os, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if sub and sub not in k: continue
        k = k[:110]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()): print("   %-28s %.4e  (%d launches)" % (name, v, cnt[(k, name)]))
