#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel dispatch (k_trace* only by default)."""
import collections
import csv
import sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
rows = list(csv.DictReader(open(path)))
per = collections.OrderedDict()
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for d in per.values() for c in d})
print("dispatch " + " ".join(names))
for k, d in per.items():
    print(k, " ".join("%.4g" % d.get(c, 0) for c in names))
