#!/usr/bin/env python3
"""Times the batched forward/inverse NTT (both passes) alone: tools/ntt_micro.py [polys] [reps] [logn]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
api = importlib.import_module("privacy-preserving-ml-through-hhe_amd.api")
polys = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
logn = int(sys.argv[3]) if len(sys.argv) > 3 else 15
Q = [1152921504595968001, 1152921504597016577, 1152921504598720513, 1152921504606584833]
lib = api.load_library(os.environ.get("HHE_LIB")) if os.environ.get("HHE_LIB") else None
if lib is not None: import torch  # same HIP runtime
X = api.Context(logn, Q, 65537, lib=lib)
n = 1 << logn
rng = np.random.default_rng(0)
h = rng.integers(0, Q[0], size=(polys, n), dtype=np.uint64)
d = torch.from_numpy(h.view(np.int64)).cuda()
for inv in (False, True):
    X.ntt(d, polys, 0, 4, inv); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        X.ntt(d, polys, 0, 4, inv)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{'inverse' if inv else 'forward'} NTT: {polys} polys N=2^{logn}: {ms:.3f} ms  {ms*1e3/polys:.3f} us/poly  {2*polys*n*8/ms/1e6:.0f} GB/s algorithmic")
