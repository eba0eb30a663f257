#!/bin/bash
# copy the summaries of tools/final_profile.sh (gpurun_out/final/) into profiles/ under this round's names
root=$(cd "$(dirname "$0")/.." && pwd)
f=$root/gpurun_out/final; p=$root/profiles
cp $f/stats/out_kernel_stats.csv $p/r3_bench_kernel_stats.csv
cp $f/pmc_summary.json $p/r3_pmc_summary.json
cp $f/bench_default.json $p/r3_bench_default.json
cp $f/bench_mnist_e2e.json $p/r3_bench_mnist_e2e.json
cp $f/c5/out_kernel_stats.csv $p/r3_config5_kernel_stats.csv
cp $f/fc/out_kernel_stats.csv $p/r3_fc_kernel_stats.csv
cp $f/c5pmc/summary.json $p/r3_config5_pmc_summary.json
python3 - <<PY
import csv, glob
out = open("$p/r3_hip_api_keyset_requests.txt", "w")
out.write("HIP API summary (rocprofv3 --hip-trace --stats) of tools/keyset_requests.py with 1 and with 4 requests on the same key sets:\n"
          "host-to-device copies per request carry the record's symmetric words and pointer tables only -- the four key objects (25 keys, 6.3 MB each) are uploaded once, before the first request.\n\n")
for r in (1, 4):
    fs = glob.glob("$f/ks%d/**/*hip_api_stats.csv" % r, recursive=True) + glob.glob("$f/ks%d/**/*hip_stats.csv" % r, recursive=True) + glob.glob("$f/ks%d/**/*domain_stats.csv" % r, recursive=True)
    out.write("== %d request(s)\n" % r)
    for f in fs[:1]:
        for row in csv.DictReader(open(f)):
            name = row.get("Name", "")
            if any(k in name for k in ("hipMemcpy", "hipMalloc", "hipFree", "hipLaunchKernel", "hipModuleLaunch")):
                out.write("  %-40s calls %8s  total %12s ns\n" % (name, row.get("Calls"), row.get("TotalDurationNs")))
    for f in glob.glob("$f/ks%d/**/*memory_copy_trace.csv" % r, recursive=True)[:1]:
        tot = {}
        for row in csv.DictReader(open(f)):
            d = row.get("Direction", "?")
            n, s = tot.get(d, (0, 0))
            tot[d] = (n + 1, s + int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for d, (n, s) in sorted(tot.items()):
            out.write("  copy-engine transfers %-28s count %6d  busy %12d ns   (this rocprofv3's trace has no size column; the count and the busy time do not grow with the requests)\n" % (d, n, s))
out.close()
print(open("$p/r3_hip_api_keyset_requests.txt").read())
PY
