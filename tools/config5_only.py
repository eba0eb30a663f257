#!/usr/bin/env python3
"""BASELINE config 5's rotation chain alone (for rocprofv3): N=2^16, 6x60-bit primes, t=8088322049, STEPS x rotate_rows(-1) over a batch.
tools/config5_only.py [steps] [batch]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
api = importlib.import_module("privacy-preserving-ml-through-hhe_amd.api")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
logn, q, t = 16, bench.Q_CONFIG5, bench.T_CONFIG5
n, K, L = 1 << logn, len(q), len(q) - 1
X = api.Context(logn, q, t)
rng = np.random.default_rng(1)
X.set_galois_key(X.query("galois_elt", -1), bench.synthetic_keys(rng, q, n))
a = torch.from_numpy(bench.synthetic_ct(rng, q, n, B).view(np.int64)).cuda()
b = torch.zeros_like(a)
X.rotate_rows(a, -1, b, B)
torch.cuda.synchronize(); t0 = time.perf_counter()
src, dst = a, b
for _ in range(steps):
    X.rotate_rows(src, -1, dst, B)
    src, dst = dst, src
X.sync(); torch.cuda.synchronize(); t1 = time.perf_counter()
per_rot = (4 * L + 2 * L * K) * 8 * n
print(f"config 5: {steps} x rotate_rows(-1), batch {B}: {1e3 * (t1 - t0) / steps:.3f} ms per rotation of the batch, {per_rot * B * steps / (t1 - t0) / 1e12:.3f} TB/s on (4L+2LK)P")
