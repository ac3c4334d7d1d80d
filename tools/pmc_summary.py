#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one directory per counter pass)."""
import csv
import glob
import sys
from collections import defaultdict


def short(n):
    return n.replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",launches")
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print(k + "," + ",".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in names) + ",%d" % n)
