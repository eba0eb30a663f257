"""Generate tests/golden/pasta_plain.json from the REFERENCE's own plain PASTA-3 code.

Runs only in the build container (needs /root/reference): `make -C oracle ref` compiles
src/pasta/pasta_3_plain.cpp + libs/keccak from source into oracle/_ref/libpasta_ref.so
(oracle/ref_shim.cpp is the extern "C" driver); this script calls it and records inputs and
expected outputs.  The fixture is data only (no reference source text).
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpasta_ref.so"))
u64p = C.POINTER(C.c_uint64)


def p(a):
    return a.ctypes.data_as(u64p)


NONCE = 123456789  # pasta_3_seal.cpp:115
cases = []
for t in (65537, 8088322049, 1096486890805657601):  # configs/config.cpp:19-26
    key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
    for block in (0, 1, 6, (1 << 32) + 5):
        mats = np.zeros((4, 2, 128, 128), np.uint64)
        rcs = np.zeros((4, 2, 128), np.uint64)
        ref.ref_pasta_block_randomness(C.c_uint64(t), C.c_uint64(NONCE), C.c_uint64(block), p(mats), p(rcs))
        ks = np.zeros(128, np.uint64)
        ref.ref_pasta_keystream(C.c_uint64(t), p(key), C.c_uint64(NONCE), C.c_uint64(block), p(ks))
        cases.append({
            "t": t, "nonce": NONCE, "block": block,
            "mats_sha256": hashlib.sha256(mats.tobytes()).hexdigest(),
            "rcs_sha256": hashlib.sha256(rcs.tobytes()).hexdigest(),
            "mat_r0_h0_row0_first8": [int(v) for v in mats[0, 0, 0, :8]],
            "mat_r3_h1_row127_last8": [int(v) for v in mats[3, 1, 127, -8:]],
            "rc_r0": [int(v) for v in rcs[0].reshape(-1)],
            "keystream": [int(v) for v in ks],
        })
enc = []
for t in (65537, 8088322049):
    key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
    for n in (1, 128, 129, 300, 784):
        pt = np.array([(7 * i + 3) % 256 for i in range(n)], dtype=np.uint64)
        ct = np.zeros(n, np.uint64)
        ref.ref_pasta_encrypt(C.c_uint64(t), p(key), p(pt), C.c_size_t(n), p(ct))
        back = np.zeros(n, np.uint64)
        ref.ref_pasta_decrypt(C.c_uint64(t), p(key), p(ct), C.c_size_t(n), p(back))
        assert (back == pt).all()
        enc.append({"t": t, "n": n, "ct": [int(v) for v in ct]})
out = {"generator": "tests/golden/make_pasta_golden.py (reference pasta_3_plain.cpp built from source)",
       "key_rule": "key[i]=(i*2654435761+12345) mod t, i<256", "pt_rule": "pt[i]=(7i+3) mod 256",
       "randomness": cases, "encrypt": enc}
path = os.path.join(ROOT, "tests", "golden", "pasta_plain.json")
json.dump(out, open(path, "w"), indent=None, separators=(",", ":"))
print("wrote", path, os.path.getsize(path), "bytes")
