#!/usr/bin/env python3
"""tests/golden/config5_chain.json: SHA-256 of the oracle's ciphertext words after 64 and 128 steps of rotate_rows(-1) at BASELINE
config 5's parameters (N = 2^16, CoeffModulus::Create(65536, {60 x 6}), t = 8088322049) on two seeded ciphertexts -- the chain the
bench times 512 steps of.  Computed once on the CPU oracle (oracle/hhe_oracle.c; minutes), compared on the GPU by
tests/test_gpu_parity.py::test_config5_rotation_chain_128_steps.  Seeds: conftest.Setup(orc, 16, [60]*6, t) and encrypt seeds 40 + b."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from conftest import Setup
orc.build()
t = 8088322049
S = Setup(orc, 16, [60] * 6, t=t)
O = S.O
rng = np.random.default_rng(5)
cts = [O.encrypt(S.pk, O.encode(rng.integers(0, 1 << 30, O.n)), 40 + b) for b in range(2)]
out = {"params": "N=65536, q=CoeffModulus::Create(65536,{60,60,60,60,60,60}), t=8088322049, Setup seeds (sk 1, pk 2, rk 3, gk 7), plaintexts default_rng(5), encrypt seeds 40+b",
       "steps": {}}
for step in range(1, 129):
    cts = [O.rotate_rows(c, -1, S.gk)[0] for c in cts]
    if step in (4, 64, 128):
        out["steps"][str(step)] = [hashlib.sha256(np.ascontiguousarray(c).tobytes()).hexdigest() for c in cts]
        print(step, out["steps"][str(step)], flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config5_chain.json"), "w"), indent=1)
