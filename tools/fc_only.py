#!/usr/bin/env python3
"""The 784x10 FC leg of bench.py's MNIST extras alone (for rocprofv3): tools/fc_only.py [samples]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
api = importlib.import_module("privacy-preserving-ml-through-hhe_amd.api")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
logn, q, t = 15, bench.Q_CONFIG2, 65537
n, K, L = 1 << logn, 4, 3
X = api.Context(logn, q, t)
rng = np.random.default_rng(1)
csp_rk, analyst_gk = X.keyset(), X.keyset()   # the key objects the CSP names at CSP.cpp:306 and :312-316
csp_rk.set_relin(bench.synthetic_keys(rng, q, n))
elts = {2 * n - 1}
g, gi = 3, pow(3, -1, 2 * n)
for _ in range(logn - 1):
    elts.add(g); elts.add(gi); g, gi = g * g % (2 * n), gi * gi % (2 * n)
for e in sorted(elts):
    analyst_gk.set_galois(e, bench.synthetic_keys(rng, q, n))
vi = torch.from_numpy(bench.synthetic_ct(rng, q, n, S * 10).view(np.int64)).cuda()
w = torch.from_numpy(bench.synthetic_ct(rng, q, n, 10).view(np.int64)).cuda()
out = torch.zeros_like(vi)
X.fc_row(vi, w, 10, 784, out, S * 10, rk=csp_rk, gk=analyst_gk)   # warm-up at the timed size: the per-kernel averages of a trace are of ONE launch shape
torch.cuda.synchronize(); t0 = time.perf_counter()
X.fc_row(vi, w, 10, 784, out, S * 10, rk=csp_rk, gk=analyst_gk)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"FC 784x10 on {S} samples: {1e3 * (t1 - t0) / S:.1f} ms/sample, fallbacks {X.query('fc_fallbacks')}")
