#!/usr/bin/env python3
"""Sum rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the dispatches of one bench run.
usage: pmc_traffic.py <fetch_dir> <write_dir> <batch> <steps_total> <out.json>
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of 16-B/lane coalesced reads, so the read side is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, sys


def load(d, name):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    per = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        per[k] += float(r["Counter_Value"])
        n[k] += 1
    return per, n


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
batch, calls = int(sys.argv[3]), int(sys.argv[4])
ours = [k for k in fetch if "at::" not in k and "rocclr" not in k]
rd = sum(fetch[k] for k in ours) * 1024 * 2
wr = sum(write.get(k, 0.0) for k in ours) * 1024
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py path kernels only",
    "correction": "read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 16-B/lane streaming reads are tallied at half); write bytes = WRITE_SIZE KiB x 1024",
    "batch": batch, "transcipher_calls": calls,
    "read_bytes_per_call": rd / calls, "write_bytes_per_call": wr / calls,
    "traffic_bytes_per_call": (rd + wr) / calls, "traffic_bytes_per_transciphering": (rd + wr) / calls / batch,
    "per_kernel_GB_per_call": {k: round((fetch[k] * 2048 + write.get(k, 0.0) * 1024) / calls / 1e9, 3) for k in sorted(ours, key=lambda k: -fetch[k])[:10]},
}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out, indent=1))
