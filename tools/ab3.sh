#!/bin/bash
# A/B of library builds incl. the per-record latency leg: tools/ab3.sh <rounds> lib...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    HHE_LIB=$lib timeout -k 10 300 python - <<PY
import json, subprocess, sys, os
p = subprocess.run([sys.executable, "bench.py", "--cpu-baseline", "0", "--extras", "0", "--steps", "3"], capture_output=True, text=True)
d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
import importlib, torch, numpy as np
import bench
api = importlib.import_module(bench.PKG + ".api")
lib = api.load_library()
D = bench.Device(torch, 0, False)
lat = bench.leg_record_latency(api, lib, D, 0)
print("$lib", round(d["value"], 1), round(d["roofline"]["avg_launch_us"], 1), "latency ms 784/300:", round(lat["784_words"]["ms_per_call"], 1), round(lat["300_words"]["ms_per_call"], 1))
PY
  done
done
