#!/usr/bin/env python3
"""tools/pmc_summary.py DIR COUNTER -- average of one rocprofv3 --pmc counter per kernel, from the
counter_collection csv files under DIR (one pass per counter, as the guide prescribes)."""
import collections
import csv
import glob
import os
import sys

d, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{counter}  calls={n:5d} avg={tot / n:16.1f}  {k[:150]}")
