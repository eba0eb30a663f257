#!/bin/bash
# PMC passes over the config-5 rotation chain (tools/config5_only.py 16 32): VALU instructions / busy / waits, fabric traffic and the clock,
# each counter set in its own run with --kernel-trace only.  Summary -> gpurun_out/<outdir>/summary.json
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o out --output-format csv -- python3 $root/tools/config5_only.py 16 32 > $out/p$i.txt 2> $out/p$i.log || echo "pass $i failed"
done
python3 - <<PY
import collections, csv, glob, json
tab = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(list)
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, f)].add(r["Dispatch_Id"])
for r in csv.DictReader(open(glob.glob("$out/p4/**/*kernel_trace.csv", recursive=True)[0])):
    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
kern = {}
for k in tab:
    if "at::" in k or "rocclr" in k or "elt_kernel" in k: continue
    n = max(len(v) for (kk, _), v in disp.items() if kk == k)
    t = tab[k]
    kern[k] = {"launches": n, "avg_us_under_the_counter_pass": sum(dur[k]) / len(dur[k]) / 1e3,
               "valu_wave_instructions_per_launch": t["SQ_INSTS_VALU"] / n, "valu_busy_quad_cycles_per_launch": t["SQ_ACTIVE_INST_VALU"] / n,
               "wave_quad_cycles_per_launch": t["SQ_WAVE_CYCLES"] / n, "wait_any_quad_cycles_per_launch": t["SQ_WAIT_ANY"] / n,
               "read_bytes_per_launch": t["FETCH_SIZE"] * 2048 / n, "write_bytes_per_launch": t["WRITE_SIZE"] * 1024 / n,
               "grbm_gui_active_per_launch": t["GRBM_GUI_ACTIVE"] / n}
out = {"source": "rocprofv3 --pmc passes of tools/config5_only.py 16 32 (N=2^16, L=5, K=6, batch 32, rotate_rows(-1) chain), tools/pmc_config5.sh",
       "correction": "read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950), write bytes = WRITE_SIZE KiB x 1024; SQ_* cycle counters are quad-cycles; GRBM_GUI_ACTIVE sums 8 XCDs",
       "kernels": kern}
json.dump(out, open("$out/summary.json", "w"), indent=1)
for k, v in kern.items():
    print("%-50s n %4d  %7.1f us  valu %.3e  busy %.0f%%  wait %.0f%%  rd %.0f MB wr %.0f MB" % (k[:50], v["launches"], v["avg_us_under_the_counter_pass"], v["valu_wave_instructions_per_launch"],
          100 * v["valu_busy_quad_cycles_per_launch"] / max(1, v["grbm_gui_active_per_launch"] / 8 * 1024 / 4) , 100 * v["wait_any_quad_cycles_per_launch"] / max(1, v["wave_quad_cycles_per_launch"]), v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
PY
