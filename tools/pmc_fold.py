#!/usr/bin/env python3
"""Mean per dispatch of every counter in rocprofv3 counter_collection.csv files, per kernel:  pmc_fold.py <csv>... [--match substr]"""
import csv, sys, collections
match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        files.append(a)
acc = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if match and match not in k:
            continue
        key = (k[:70], row["Grid_Size"], row["Counter_Name"])
        a = acc[key]; a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, g, c), (s, n) in sorted(acc.items()):
    print("%-70s grid %-9s %-34s %14.1f  (x%d)" % (k, g, c, s / n, n))
