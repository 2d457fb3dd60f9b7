#!/usr/bin/env python3
"""per-kernel means of every counter found under a rocprofv3 --pmc output directory: python tools/pmc_dump.py DIR [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
root, match = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if match and match not in k:
            continue
        c = acc[k][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"]); c[1] += 1
for k, cs in acc.items():
    print(k[:120])
    for name, (s, n) in sorted(cs.items()):
        print(f"   {name:28s} mean {s / n:16.1f}  (n={n})")
