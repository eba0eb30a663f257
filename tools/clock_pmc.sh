#!/bin/bash
# shader clock per kernel during the bench workload: GRBM_GUI_ACTIVE (one PMC pass, --kernel-trace only) / 8 XCDs / duration.  tools/clock_pmc.sh lib.so ...
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=$root/gpurun_out/clockpmc_$tag
  export HHE_LIB=$root/$lib
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $out -o out --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 --kernel-timing 0 > $out.json 2> $out.log )
  python3 - <<PY
import csv, collections
dur = {}
for r in csv.DictReader(open("$out/out_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open("$out/out_counter_collection.csv")):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, d = dur[r["Dispatch_Id"]]
    a = acc[name.split("(")[0]]
    a[0] += 1; a[1] += d; a[2] += float(r["Counter_Value"])
print("== $lib")
for k, (n, d, c) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:4]:
    print("  %-58s launches %5d  avg %7.1f us  %.3f GHz" % (k[:58], n, d / n / 1e3, c / 8 / d))
PY
done
