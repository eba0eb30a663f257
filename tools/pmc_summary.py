#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc passes: tools/pmc_summary.py <dir with p*/> [kernel filter]"""
import collections, csv, glob, re, sys
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
tab = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES" or r["Counter_Name"] == "FETCH_SIZE":
            seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen:
        calls[k] = max(calls[k], sum(1 for kk, _ in seen if kk == k))
for k in sorted(tab, key=lambda k: -tab[k].get("SQ_WAVE_CYCLES", 0)):
    if not re.search(flt, k):
        continue
    print(k, "dispatches", calls[k])
    for c in sorted(tab[k]):
        print(f"    {c:28s} {tab[k][c]:.4g}   per dispatch {tab[k][c] / max(1, calls[k]):.4g}")
