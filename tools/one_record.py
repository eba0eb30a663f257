#!/usr/bin/env python3
"""hhe_decompose of ONE 784-word record, CALLS times (the drop-in's call shape, CSP.cpp:247-252) for `rocprofv3 --kernel-trace --stats`:
the sum of the kernel durations against the wall time says whether a one-record call is bound by the GPU's per-launch latency chain or by
the host's launch rate.  tools/one_record.py CALLS [words]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
api = importlib.import_module("privacy-preserving-ml-through-hhe_amd.api")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
words = int(sys.argv[2]) if len(sys.argv) > 2 else 784
logn, q, t = 15, bench.Q_CONFIG2, 65537
n, K, L = 1 << logn, 4, 3
X = api.Context(logn, q, t)
rng = np.random.default_rng(1)
X.set_relin_key(bench.synthetic_keys(rng, q, n))
for e in sorted({X.query("galois_elt", s) for s in [0, -1, 128] + [-128 * i for i in range(1, 7)]}):
    X.set_galois_key(e, bench.synthetic_keys(rng, q, n))
enc_key = torch.from_numpy(bench.synthetic_ct(rng, q, n).view(np.int64)).cuda()
recs = rng.integers(0, t, size=(1, words), dtype=np.uint64)
flat = torch.zeros((1, 2, L, n), dtype=torch.int64, device="cuda")
X.decompose(enc_key, recs, flat)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(calls):
    X.decompose(enc_key, recs, flat)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one {words}-word record per call: {1e3 * (t1 - t0) / calls:.2f} ms per call")
