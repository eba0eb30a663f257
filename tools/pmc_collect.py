#!/usr/bin/env python3
"""Summarise tools/pmc_passes.sh output into the JSON bench.py reads (profiles/r3_pmc_summary.json).
usage: pmc_collect.py <pmc dir> <batch> <transcipher calls> <out.json>
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of 16-B/lane coalesced reads, so the read side is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

d, batch, calls, outp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, f)].add(r["Dispatch_Id"])
ndisp = {k: max(len(v) for (kk, _), v in disp.items() if kk == k) for k in tab}
ours = [k for k in tab if "at::" not in k and "rocclr" not in k]
rd = sum(tab[k].get("FETCH_SIZE", 0) for k in ours) * 1024 * 2
wr = sum(tab[k].get("WRITE_SIZE", 0) for k in ours) * 1024
valu = sum(tab[k].get("SQ_INSTS_VALU", 0) for k in ours)
kern = {}
for k in sorted(ours, key=lambda k: -(tab[k].get("FETCH_SIZE", 0) * 2 + tab[k].get("WRITE_SIZE", 0)))[:12]:
    n = ndisp[k]
    name = k.split("<")[0] if k.startswith("ks_row_kernel") else k
    kern[name] = {"launches": n,
                  "traffic_bytes_per_launch": (tab[k].get("FETCH_SIZE", 0) * 2048 + tab[k].get("WRITE_SIZE", 0) * 1024) / n,
                  "read_bytes_per_launch": tab[k].get("FETCH_SIZE", 0) * 2048 / n, "write_bytes_per_launch": tab[k].get("WRITE_SIZE", 0) * 1024 / n,
                  "valu_wave_instructions_per_launch": tab[k].get("SQ_INSTS_VALU", 0) / n,
                  "valu_busy_quad_cycles_per_launch": tab[k].get("SQ_ACTIVE_INST_VALU", 0) / n,
                  "wait_any_quad_cycles_per_launch": tab[k].get("SQ_WAIT_ANY", 0) / n,
                  "wait_inst_any_quad_cycles_per_launch": tab[k].get("SQ_WAIT_INST_ANY", 0) / n,
                  "wave_quad_cycles_per_launch": tab[k].get("SQ_WAVE_CYCLES", 0) / n,
                  "grbm_gui_active_per_launch": tab[k].get("GRBM_GUI_ACTIVE", 0) / n,
                  "tcc_hit_per_launch": tab[k].get("TCC_HIT_sum", 0) / n, "tcc_miss_per_launch": tab[k].get("TCC_MISS_sum", 0) / n,
                  "lds_bank_conflict_cycles_per_launch": tab[k].get("SQ_LDS_BANK_CONFLICT", 0) / n}
out = {"source": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --kernel-timing 0 --batch %d` (tools/pmc_passes.sh), path kernels only" % batch,
       "source_hash": bench.source_hash(),
       "correction": "read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950: 16-B/lane streaming reads are tallied at half); write bytes = WRITE_SIZE KiB x 1024; SQ_* cycle counters are quad-cycles",
       "batch": batch, "transcipher_calls": calls,
       "traffic_bytes_per_transciphering": (rd + wr) / calls / batch, "read_bytes_per_transciphering": rd / calls / batch,
       "write_bytes_per_transciphering": wr / calls / batch,
       "valu_wave_instructions_per_transciphering": valu / calls / batch, "kernels": kern}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("source_hash", "traffic_bytes_per_transciphering", "valu_wave_instructions_per_transciphering")}, indent=1))
