"""ctypes binding of oracle/libhhe_oracle.so (CPU oracle; test infrastructure only).

The C file is the restatement; this module only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhhe_oracle.so")
PASTA_T = 128
PASTA_NONCE = 123456789


def build(force=False):
    src = os.path.join(_HERE, "hhe_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libhhe_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


class _GK(C.Structure):
    _fields_ = [("nk", C.c_int), ("elts", u32p), ("keys", u64p)]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_ctx_create.restype = C.c_void_p
        _lib.orc_ctx_create.argtypes = [C.c_int, C.c_int, u64p, C.c_uint64]
        _lib.orc_minimal_primitive_root.restype = C.c_uint64
        _lib.orc_minimal_primitive_root.argtypes = [C.c_uint64, C.c_uint64]
        _lib.orc_ctx_query.restype = C.c_uint64
        _lib.orc_ctx_query.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        _lib.orc_galois_elt_from_step.restype = C.c_uint32
        _lib.orc_is_prime.argtypes = [C.c_uint64]
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def u64(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def get_primes(factor, bits, count):
    out = np.zeros(count, np.uint64)
    rc = lib().orc_get_primes(C.c_uint64(factor), C.c_int(bits), C.c_size_t(count), _p(out))
    assert rc == 0
    return [int(v) for v in out]


def coeff_modulus_create(n, bit_sizes):
    out = np.zeros(len(bit_sizes), np.uint64)
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    rc = lib().orc_coeff_modulus_create(C.c_size_t(n), bs, C.c_size_t(len(bit_sizes)), _p(out))
    assert rc == 0
    return [int(v) for v in out]


def minimal_primitive_root(degree, q):
    return int(lib().orc_minimal_primitive_root(degree, q))


def naf(v):
    out = (C.c_int * 40)()
    n = lib().orc_naf(C.c_int(v), out)
    return [out[i] for i in range(n)]


def shake128(data: bytes, outlen: int) -> bytes:
    st = C.create_string_buffer(25 * 8 + 168 + 8)
    buf = C.create_string_buffer(outlen)
    lib().orc_shake128_init(st, data, C.c_size_t(len(data)))
    lib().orc_shake128_squeeze(st, buf, C.c_size_t(outlen))
    return buf.raw


def pasta_block_randomness(t, block, nonce=PASTA_NONCE):
    mats = np.zeros((4, 2, PASTA_T, PASTA_T), np.uint64)
    rcs = np.zeros((4, 2, PASTA_T), np.uint64)
    lib().orc_pasta_block_randomness(C.c_uint64(t), C.c_uint64(nonce), C.c_uint64(block), _p(mats), _p(rcs))
    return mats, rcs


def pasta_keystream(t, key, block, nonce=PASTA_NONCE):
    ks = np.zeros(PASTA_T, np.uint64)
    lib().orc_pasta_keystream(C.c_uint64(t), _p(u64(key)), C.c_uint64(nonce), C.c_uint64(block), _p(ks))
    return ks


def pasta_encrypt(t, key, pt):
    pt = u64(pt)
    ct = np.zeros_like(pt)
    lib().orc_pasta_encrypt(C.c_uint64(t), _p(u64(key)), _p(pt), C.c_size_t(len(pt)), _p(ct))
    return ct


def pasta_decrypt(t, key, ct):
    ct = u64(ct)
    pt = np.zeros_like(ct)
    lib().orc_pasta_decrypt(C.c_uint64(t), _p(u64(key)), _p(ct), C.c_size_t(len(ct)), _p(pt))
    return pt


class GaloisKeys:
    """elts[nk] + keys[nk][L][2][K][N] (NTT form), the layout of KSwitchKeys::data()."""

    def __init__(self, elts, keys):
        self.elts = np.ascontiguousarray(elts, dtype=np.uint32)
        self.keys = u64(keys)
        self.c = _GK(len(self.elts), self.elts.ctypes.data_as(u32p), _p(self.keys))

    def ref(self):
        return C.byref(self.c)


class Oracle:
    def __init__(self, logn, q, t):
        self.logn, self.n, self.q, self.t = logn, 1 << logn, [int(v) for v in q], int(t)
        self.K, self.L = len(q), len(q) - 1
        qa = u64(self.q)
        self.h = C.c_void_p(lib().orc_ctx_create(logn, self.K, _p(qa), C.c_uint64(t)))
        assert self.h.value, "orc_ctx_create failed"
        self.ct_shape = (2, self.L, self.n)
        self.ksk_shape = (self.L, 2, self.K, self.n)

    def __del__(self):
        try:
            lib().orc_ctx_destroy(self.h)
        except Exception:
            pass

    def query(self, what, i=0):
        return int(lib().orc_ctx_query(self.h, what.encode(), i))

    def ntt_table(self, mi, inverse=False, shoup=False):
        out = np.zeros(self.n, np.uint64)
        lib().orc_ctx_ntt_table(self.h, C.c_int(mi), int(inverse), int(shoup), _p(out))
        return out

    def ntt_fwd(self, mi, a):
        a = u64(a).copy()
        lib().orc_ntt_fwd(self.h, C.c_int(mi), _p(a))
        return a

    def ntt_inv(self, mi, a):
        a = u64(a).copy()
        lib().orc_ntt_inv(self.h, C.c_int(mi), _p(a))
        return a

    def encode(self, vals):
        vals = u64(np.asarray(vals, dtype=np.int64) % self.t)
        out = np.zeros(self.n, np.uint64)
        lib().orc_encode(self.h, _p(vals), C.c_size_t(len(vals)), _p(out))
        return out

    def decode(self, plain):
        out = np.zeros(self.n, np.uint64)
        lib().orc_decode(self.h, _p(u64(plain)), _p(out))
        return out

    def galois_elt(self, step):
        return int(lib().orc_galois_elt_from_step(self.h, C.c_int(step)))

    def galois_elts_all(self):
        out = (C.c_uint32 * 64)()
        n = lib().orc_galois_elts_all(self.h, out)
        return [out[i] for i in range(n)]

    def galois_poly(self, mi, elt, a):
        out = np.zeros(self.n, np.uint64)
        lib().orc_apply_galois_poly(self.h, C.c_int(mi), C.c_uint32(elt), _p(u64(a)), _p(out))
        return out

    # --- keys ---
    def keygen_secret(self, seed):
        sk = np.zeros((self.K, self.n), np.uint64)
        lib().orc_keygen_secret(self.h, C.c_uint64(seed), _p(sk))
        return sk

    def keygen_public(self, sk, seed):
        pk = np.zeros((2, self.K, self.n), np.uint64)
        lib().orc_keygen_public(self.h, _p(sk), C.c_uint64(seed), _p(pk))
        return pk

    def keygen_relin(self, sk, seed):
        k = np.zeros(self.ksk_shape, np.uint64)
        lib().orc_keygen_relin(self.h, _p(sk), C.c_uint64(seed), _p(k))
        return k

    def keygen_galois(self, sk, elts, seed):
        keys = np.zeros((len(elts),) + self.ksk_shape, np.uint64)
        for i, e in enumerate(elts):
            lib().orc_keygen_galois(self.h, _p(sk), C.c_uint32(e), C.c_uint64(seed + 1000 * (i + 1)), _p(keys[i]))
        return GaloisKeys(elts, keys)

    def encrypt(self, pk, plain, seed):
        ct = np.zeros(self.ct_shape, np.uint64)
        lib().orc_encrypt(self.h, _p(pk), _p(u64(plain)), C.c_uint64(seed), _p(ct))
        return ct

    def encrypt_symmetric(self, sk, plain, seed):
        ct = np.zeros(self.ct_shape, np.uint64)
        lib().orc_encrypt_symmetric(self.h, _p(sk), _p(u64(plain)), C.c_uint64(seed), _p(ct))
        return ct

    def decrypt(self, sk, ct):
        ct = u64(ct)
        out = np.zeros(self.n, np.uint64)
        lib().orc_decrypt(self.h, _p(sk), _p(ct), C.c_int(ct.shape[0]), _p(out))
        return out

    def phase(self, sk, ct):
        ct = u64(ct)
        out = np.zeros((self.L, self.n), np.uint64)
        lib().orc_phase(self.h, _p(sk), _p(ct), C.c_int(ct.shape[0]), _p(out))
        return out

    def noise_budget(self, sk, ct, sample=64):
        """invariant noise budget (bits) from `sample` coefficients, exact bigint CRT (tests only)."""
        ph = self.phase(sk, ct)
        qs = self.q[: self.L]
        Q = 1
        for v in qs:
            Q *= v
        worst = 0
        idx = np.linspace(0, self.n - 1, min(sample, self.n)).astype(int)
        for i in idx:
            x = 0
            for j, qj in enumerate(qs):
                Mj = Q // qj
                x += int(ph[j, i]) * Mj * pow(Mj, -1, qj)
            x %= Q
            v = (x * self.t) % Q
            if v > Q // 2:
                v = Q - v
            worst = max(worst, v)
        if worst == 0:
            return Q.bit_length()
        return (Q // (2 * worst)).bit_length() - 1

    # --- evaluator ---
    def add(self, a, b):
        a, b = u64(a), u64(b)
        out = np.zeros_like(a)
        lib().orc_add(self.h, _p(a), _p(b), C.c_int(a.shape[0]), _p(out))
        return out

    def negate(self, a):
        a = u64(a)
        out = np.zeros_like(a)
        lib().orc_negate(self.h, _p(a), C.c_int(a.shape[0]), _p(out))
        return out

    def add_plain(self, a, plain):
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_add_plain(self.h, _p(u64(a)), _p(u64(plain)), _p(out))
        return out

    def sub_plain(self, a, plain):
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_sub_plain(self.h, _p(u64(a)), _p(u64(plain)), _p(out))
        return out

    def multiply_plain(self, a, plain):
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_multiply_plain(self.h, _p(u64(a)), _p(u64(plain)), _p(out))
        return out

    def switch_key(self, ct, d, ksk):
        ct = u64(ct).copy()
        lib().orc_switch_key(self.h, _p(ct), _p(u64(d)), _p(u64(ksk)))
        return ct

    def apply_galois(self, a, elt, ksk):
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_apply_galois(self.h, _p(u64(a)), C.c_uint32(elt), _p(u64(ksk)), _p(out))
        return out

    def rotate_rows(self, a, step, gk):
        out = np.zeros(self.ct_shape, np.uint64)
        r = lib().orc_rotate_rows(self.h, _p(u64(a)), C.c_int(step), gk.ref(), _p(out))
        assert r >= 0, "Galois key not present"
        return out, r

    def rotate_columns(self, a, gk):
        out = np.zeros(self.ct_shape, np.uint64)
        r = lib().orc_rotate_columns(self.h, _p(u64(a)), gk.ref(), _p(out))
        assert r >= 0
        return out

    def multiply(self, a, b):
        out = np.zeros((3, self.L, self.n), np.uint64)
        lib().orc_multiply(self.h, _p(u64(a)), _p(u64(b)), _p(out))
        return out

    def relinearize(self, a3, rk):
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_relinearize(self.h, _p(u64(a3)), _p(u64(rk)), _p(out))
        return out

    # --- hot path ---
    def pasta_pack_key(self, key256):
        out = np.zeros(self.n, np.uint64)
        lib().orc_pasta_pack_key(self.h, _p(u64(key256)), _p(out))
        return out

    def transcipher_block(self, enc_key, rk, gk, cw, block_index, use_bsgs=False):
        cw = u64(cw)
        out = np.zeros(self.ct_shape, np.uint64)
        rc = lib().orc_pasta_transcipher_block(self.h, _p(u64(enc_key)), _p(u64(rk)), gk.ref(), _p(cw),
                                               C.c_size_t(len(cw)), C.c_uint64(block_index), int(use_bsgs), _p(out))
        if rc == -2:
            raise RuntimeError("too little slots for matmul implementation!")
        assert rc == 0, "Galois key not present"
        return out

    def transcipher_batch(self, enc_key, rk, gk, cw, ncw, block_index, threads=1):
        cw = u64(cw)
        nb = cw.shape[0]
        assert cw.shape[1] == PASTA_T
        ncw = np.ascontiguousarray(ncw, dtype=np.uint32)
        bi = u64(block_index)
        out = np.zeros((nb,) + self.ct_shape, np.uint64)
        rc = lib().orc_pasta_transcipher_batch(self.h, _p(u64(enc_key)), _p(u64(rk)), gk.ref(), _p(cw),
                                               ncw.ctypes.data_as(u32p), _p(bi), C.c_size_t(nb), C.c_int(threads), _p(out))
        assert rc == 0
        return out

    def mask(self, a, mask_vals):
        mv = u64(mask_vals)
        out = np.zeros(self.ct_shape, np.uint64)
        lib().orc_mask(self.h, _p(u64(a)), _p(mv), C.c_size_t(len(mv)), _p(out))
        return out

    def flatten(self, blocks, gk):
        blocks = u64(blocks)
        out = np.zeros(self.ct_shape, np.uint64)
        rc = lib().orc_flatten(self.h, _p(blocks), C.c_size_t(blocks.shape[0]), gk.ref(), _p(out))
        assert rc == 0
        return out

    def fc_row(self, vi, w, rk, gk, n_inputs):
        out = np.zeros(self.ct_shape, np.uint64)
        ks = lib().orc_fc_row(self.h, _p(u64(vi)), _p(u64(w)), _p(u64(rk)), gk.ref(), C.c_size_t(n_inputs), _p(out))
        assert ks >= 0
        return out, ks
