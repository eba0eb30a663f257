#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel summary: tools/kstats.py <out_kernel_stats.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:n]:
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):6.2f} %')
