#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV directory: mean counter value per kernel per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
