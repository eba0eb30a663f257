#!/usr/bin/env python3
"""Builds tests/golden/ecg_fc1.json from DATA files of the reference (run in the build container only):
  weights/ecg/ecg_512/fc1_weight_50epochs_bz4.csv   128 x 1 integer weights of the ECG model's first layer
  weights/ecg/ecg_512/fc1_bias_50epochs_bs4.csv     its bias
(BASELINE config 4; the ECG inputs data/mit-bih/csv/mitbih_x_test_int.csv are missing from the reference, SURVEY 8d, so the
tests pair these weights with seeded synthetic 128-word inputs in [0,255].)  Only numbers are copied; no reference code."""
import json, os
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rd = lambda p: [int(v) for line in open(os.path.join(REF, p)) for v in line.strip().rstrip(",").split(",") if v.strip()]
w = rd("weights/ecg/ecg_512/fc1_weight_50epochs_bz4.csv")
b = rd("weights/ecg/ecg_512/fc1_bias_50epochs_bs4.csv")
assert len(w) == 128 and len(b) == 1
out = {"source": "tests/golden/make_ecg_fixture.py (reference data files weights/ecg/ecg_512/fc1_{weight,bias}_50epochs_*.csv)",
       "fc1_weight": w, "fc1_bias": b}
path = os.path.join(ROOT, "tests", "golden", "ecg_fc1.json")
json.dump(out, open(path, "w"), separators=(",", ":"))
print("wrote", path, "weights", len(w), "min/max", min(w), max(w), "bias", b)
