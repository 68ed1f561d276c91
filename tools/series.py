#!/usr/bin/env python3
"""usage: tools/series.py <kernel_trace.csv> [substring]  -- per-launch durations (us) of one kernel in launch order;
long series are printed as means over groups of launches"""
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else "sad_surface"
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = 1 if len(d) <= 80 else (len(d) + 59) // 60
print(f"{len(d)} launches; us (means of {g}):", " ".join(f"{sum(d[i:i + g]) / len(d[i:i + g]):.0f}" for i in range(0, len(d), g)))
