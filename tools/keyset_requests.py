#!/usr/bin/env python3
"""REQUESTS CSP requests on one context, each naming the reference's key objects per call (CSP.cpp:238-242, 271-278, 306, 312-316) as key
sets: decompose (PASTA keys + flatten keys) of one 300-word record, then one FC row (CSP relin key + the analyst's default Galois keys).
The key sets are created before the first request (what the adapters' content-hash cache does on first sight of an object).  For
`rocprofv3 --hip-trace --stats`: host-to-device copies per request must not include key material -- compare REQUESTS = 1 and 4.
tools/keyset_requests.py REQUESTS"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
api = importlib.import_module("privacy-preserving-ml-through-hhe_amd.api")
reqs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = api.load_library()
D = bench.Device(torch, 0, False)
F = bench.MnistFlow(api, lib, D, 0, 15, bench.Q_CONFIG2, bench.T_PLAIN, nin=300, neurons=1)   # uploads the four key objects once
rec = F.records(0, 1)
for r in range(reqs):
    F.run(rec)
torch.cuda.synchronize()
print("requests", reqs, "done; key objects resident: 4 sets")
F.close()
