"""ctypes binding of libhhe_gfx950.so (include/hhe_gfx950.h).

The library is the product: hand-written gfx950 kernels behind a C ABI.  This module
only marshals pointers; device memory comes from PyTorch-ROCm tensors (plumbing).
There is no CPU path: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhhe_gfx950.so")
PASTA_T = 128

u64p = C.POINTER(C.c_uint64)


class HheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hhe error {code}: {msg}")
        self.code = code


# error codes of include/hhe_gfx950.h
ERR_INVALID, ERR_NO_GALOIS_KEY, ERR_TOO_FEW_SLOTS, ERR_DEVICE, ERR_NO_RELIN_KEY = 1, 2, 3, 4, 5

_SYMBOLS = [
    "hhe_last_error", "hhe_backend", "hhe_ctx_create", "hhe_bfv_default_coeff_modulus", "hhe_ctx_destroy", "hhe_ctx_set_stream",
    "hhe_ctx_reserve", "hhe_ctx_sync", "hhe_ctx_query", "hhe_set_relin_key", "hhe_set_relin_key_slot", "hhe_set_galois_key",
    "hhe_has_galois_key", "hhe_malloc", "hhe_free", "hhe_copy_h2d", "hhe_copy_d2h", "hhe_ntt",
    "hhe_encode", "hhe_add", "hhe_negate", "hhe_add_plain", "hhe_multiply_plain", "hhe_apply_galois",
    "hhe_rotate_rows", "hhe_rotate_columns", "hhe_multiply", "hhe_relinearize",
    "hhe_pasta3_transcipher", "hhe_pasta3_clear_block_cache", "hhe_mask", "hhe_flatten", "hhe_fc_row", "hhe_decompose",
    "hhe_pasta3_block_randomness", "hhe_pasta3_plain_keystream", "hhe_pasta3_plain_crypt", "hhe_decrypt",
    "hhe_ctx_profile", "hhe_ctx_profile_read", "hhe_relinearize_slot",
    "hhe_seal_load_ciphertext", "hhe_seal_save_ciphertext", "hhe_seal_load_relin_keys", "hhe_seal_load_galois_keys",
    "hhe_pasta3_set_block_cache_limit", "hhe_keyset_create", "hhe_keyset_destroy", "hhe_keyset_set_relin", "hhe_keyset_set_galois", "hhe_keyset_has_galois", "hhe_keyset_has_relin",
    "hhe_apply_galois_ks", "hhe_rotate_rows_ks", "hhe_rotate_columns_ks", "hhe_relinearize_ks", "hhe_pasta3_transcipher_ks",
    "hhe_flatten_ks", "hhe_decompose_ks", "hhe_fc_row_ks", "hhe_seal_load_relin_keys_ks", "hhe_seal_load_galois_keys_ks",
]


def exported_symbols():
    return list(_SYMBOLS)


def load_library(path=None):
    """Load the C-ABI library.  Default: the in-tree gfx950 build; anything else must be passed explicitly."""
    path = path or os.environ.get("HHE_LIB") or LIB_PATH  # HHE_LIB: another gfx950 build of the same sources (A/B runs)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the gfx950 HIP library is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if os.path.abspath(path) == os.path.abspath(LIB_PATH) or path == os.environ.get("HHE_LIB"):
        # Device memory and streams are shared with PyTorch-ROCm, so both must sit on ONE HIP runtime:
        # import torch first so libamdhip64.so.7 resolves to the copy torch already loaded (two HSA
        # runtimes in one process cannot both open the GPU).
        import torch  # noqa: F401
    lib = C.CDLL(path)
    for s in _SYMBOLS:
        getattr(lib, s)  # AttributeError if the ABI is incomplete
    lib.hhe_last_error.restype = C.c_char_p
    lib.hhe_backend.restype = C.c_char_p
    lib.hhe_ctx_query.restype = C.c_uint64
    lib.hhe_ctx_query.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.hhe_malloc.restype = C.c_void_p
    lib.hhe_malloc.argtypes = [C.c_size_t]
    lib.hhe_free.argtypes = [C.c_void_p]
    lib.hhe_ctx_destroy.argtypes = [C.c_void_p]
    lib.hhe_pasta3_clear_block_cache.argtypes = [C.c_void_p]
    lib.hhe_keyset_destroy.argtypes = [C.c_void_p]
    lib.hhe_keyset_destroy.restype = None
    return lib


def _ptr(x):
    """address of a torch tensor / numpy array / raw int"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return C.c_void_p(x.ctypes.data)
    assert x.is_contiguous()
    return C.c_void_p(x.data_ptr())


def _ks(keyset):
    """handle of a KeySet, or NULL = the context's default set"""
    return keyset.h if keyset is not None else C.c_void_p(0)


class KeySet:
    """One seal::RelinKeys / seal::GaloisKeys object on the device (hhe_keyset): keys with identity."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx.lib.hhe_keyset_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.hhe_keyset_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_relin(self, ksk):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        assert ksk.shape == (self.ctx.L, 2, self.ctx.K, self.ctx.n)
        self.ctx._chk(self.ctx.lib.hhe_keyset_set_relin(self.h, _ptr(ksk)))
        return self

    def set_galois(self, elt, ksk):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        assert ksk.shape == (self.ctx.L, 2, self.ctx.K, self.ctx.n)
        self.ctx._chk(self.ctx.lib.hhe_keyset_set_galois(self.h, C.c_uint32(elt), _ptr(ksk)))
        return self

    def has_galois(self, elt):
        return bool(self.ctx.lib.hhe_keyset_has_galois(self.h, C.c_uint32(elt)))

    def has_relin(self):
        return bool(self.ctx.lib.hhe_keyset_has_relin(self.h))

    def seal_load_relin_keys(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
        used = C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.hhe_seal_load_relin_keys_ks(self.h, buf, C.c_size_t(len(blob)), C.byref(used)))
        return int(used.value)

    def seal_load_galois_keys(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
        used, cnt = C.c_size_t(0), C.c_uint32(0)
        self.ctx._chk(self.ctx.lib.hhe_seal_load_galois_keys_ks(self.h, buf, C.c_size_t(len(blob)), C.byref(used), C.byref(cnt)))
        return int(used.value), int(cnt.value)


class Context:
    """One BFV context on one GPU (hhe_ctx)."""

    def __init__(self, logn, q, t, device=0, lib=None):
        self.lib = lib or load_library()
        self.logn, self.n, self.q, self.t = logn, 1 << logn, [int(v) for v in q], int(t)
        self.K, self.L = len(q), len(q) - 1
        qa = np.asarray(self.q, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(self.lib.hhe_ctx_create(C.c_int(logn), C.c_int(self.K), _ptr(qa), C.c_uint64(t), C.c_int(device), C.byref(h)))
        self.h = h
        self.ct_shape = (2, self.L, self.n)

    def _chk(self, rc):
        if rc != 0:
            raise HheError(rc, self.lib.hhe_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.hhe_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def backend(self):
        return self.lib.hhe_backend().decode()

    def query(self, what, i=0):
        return int(self.lib.hhe_ctx_query(self.h, what.encode(), C.c_int(i)))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.hhe_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def reserve(self, B):
        self._chk(self.lib.hhe_ctx_reserve(self.h, C.c_size_t(B)))

    def sync(self):
        self._chk(self.lib.hhe_ctx_sync(self.h))

    def profile(self, enable=True):
        """bracket every launch of the fused key-switch row kernel with timed HIP events on its stream"""
        self._chk(self.lib.hhe_ctx_profile(self.h, C.c_int(1 if enable else 0)))

    def profile_read(self):
        """(kernel name as rocprofv3 prints it, launches, total ms, ciphertexts covered) since the last read"""
        name = C.create_string_buffer(128)
        n, items, ms = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        self._chk(self.lib.hhe_ctx_profile_read(self.h, name, C.c_size_t(128), C.byref(n), C.byref(ms), C.byref(items)))
        return name.value.decode(), int(n.value), float(ms.value), int(items.value)

    def set_relin_key(self, ksk):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        assert ksk.shape == (self.L, 2, self.K, self.n)
        self._chk(self.lib.hhe_set_relin_key(self.h, _ptr(ksk)))

    def set_galois_key(self, elt, ksk):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        assert ksk.shape == (self.L, 2, self.K, self.n)
        self._chk(self.lib.hhe_set_galois_key(self.h, C.c_uint32(elt), _ptr(ksk)))

    def has_galois_key(self, elt):
        return bool(self.lib.hhe_has_galois_key(self.h, C.c_uint32(elt)))

    # ---- primitives on device batches (arguments: torch cuda tensors / raw addresses) ----
    def ntt(self, polys, count, mod_base, mod_cycle, inverse=False):
        self._chk(self.lib.hhe_ntt(self.h, _ptr(polys), C.c_size_t(count), C.c_int(mod_base), C.c_int(mod_cycle), C.c_int(int(inverse))))

    def encode(self, vals, B, count, plain):
        self._chk(self.lib.hhe_encode(self.h, _ptr(vals), C.c_size_t(B), C.c_size_t(count), _ptr(plain)))

    def add(self, a, b, out, B, size=2):
        self._chk(self.lib.hhe_add(self.h, _ptr(a), _ptr(b), _ptr(out), C.c_size_t(B), C.c_int(size)))

    def negate(self, a, out, B, size=2):
        self._chk(self.lib.hhe_negate(self.h, _ptr(a), _ptr(out), C.c_size_t(B), C.c_int(size)))

    def add_plain(self, ct, plain, out, B, bcast=False, subtract=False):
        self._chk(self.lib.hhe_add_plain(self.h, _ptr(ct), _ptr(plain), C.c_int(int(bcast)), C.c_int(int(subtract)), _ptr(out), C.c_size_t(B)))

    def multiply_plain(self, ct, plain, out, B, bcast=False):
        self._chk(self.lib.hhe_multiply_plain(self.h, _ptr(ct), _ptr(plain), C.c_int(int(bcast)), _ptr(out), C.c_size_t(B)))

    def keyset(self):
        return KeySet(self)

    # gk / rk: the KeySet the call names (None = the context's default set)
    def apply_galois(self, ct, elt, out, B, gk=None):
        self._chk(self.lib.hhe_apply_galois_ks(self.h, _ks(gk), _ptr(ct), C.c_uint32(elt), _ptr(out), C.c_size_t(B)))

    def rotate_rows(self, ct, step, out, B, gk=None):
        self._chk(self.lib.hhe_rotate_rows_ks(self.h, _ks(gk), _ptr(ct), C.c_int(step), _ptr(out), C.c_size_t(B)))

    def rotate_columns(self, ct, out, B, gk=None):
        self._chk(self.lib.hhe_rotate_columns_ks(self.h, _ks(gk), _ptr(ct), _ptr(out), C.c_size_t(B)))

    def multiply(self, a, b, out3, B):
        self._chk(self.lib.hhe_multiply(self.h, _ptr(a), _ptr(b), _ptr(out3), C.c_size_t(B)))

    def relinearize(self, a3, out, B, slot=None, rk=None):
        if rk is not None:
            self._chk(self.lib.hhe_relinearize_ks(self.h, _ks(rk), _ptr(a3), _ptr(out), C.c_size_t(B)))
        elif slot is None:
            self._chk(self.lib.hhe_relinearize(self.h, _ptr(a3), _ptr(out), C.c_size_t(B)))
        else:
            self._chk(self.lib.hhe_relinearize_slot(self.h, C.c_int(slot), _ptr(a3), _ptr(out), C.c_size_t(B)))

    # ---- hot path ----
    def transcipher(self, enc_key, cw, ncw, block_index, out, use_bsgs=False, rk=None, gk=None):
        """cw: host uint64 [B][128]; ncw [B]; block_index [B]; enc_key/out device."""
        cw = np.ascontiguousarray(cw, dtype=np.uint64)
        B = cw.shape[0]
        assert cw.shape == (B, PASTA_T)
        ncw = np.ascontiguousarray(ncw, dtype=np.uint32)
        bi = np.ascontiguousarray(block_index, dtype=np.uint64)
        assert ncw.shape == (B,) and bi.shape == (B,)
        self._chk(self.lib.hhe_pasta3_transcipher_ks(self.h, _ks(rk), _ks(gk), _ptr(enc_key), _ptr(cw), _ptr(ncw), _ptr(bi), C.c_size_t(B),
                                                     C.c_int(int(use_bsgs)), _ptr(out)))

    def clear_block_cache(self):
        self.lib.hhe_pasta3_clear_block_cache(self.h)

    def set_block_cache_limit(self, nbytes):
        self._chk(self.lib.hhe_pasta3_set_block_cache_limit(self.h, C.c_size_t(nbytes)))

    def mask(self, ct, mask_vals, out, B):
        mv = np.ascontiguousarray(mask_vals, dtype=np.uint64)
        self._chk(self.lib.hhe_mask(self.h, _ptr(ct), _ptr(mv), C.c_size_t(len(mv)), _ptr(out), C.c_size_t(B)))

    def flatten(self, blocks, nblocks, out, S, gk=None):
        self._chk(self.lib.hhe_flatten_ks(self.h, _ks(gk), _ptr(blocks), C.c_size_t(nblocks), _ptr(out), C.c_size_t(S)))

    def decompose(self, enc_key, records, out, mask_last=True, rk=None, gk=None, flatten_gk=None):
        """records: host uint64 [S][nwords]; out device [S][2][L][N]; rk / gk: the PASTA_SEAL's keys, flatten_gk: the GaloisKeys of flatten"""
        rec = np.ascontiguousarray(records, dtype=np.uint64)
        S, nwords = rec.shape
        self._chk(self.lib.hhe_decompose_ks(self.h, _ks(rk), _ks(gk), _ks(flatten_gk), _ptr(enc_key), _ptr(rec), C.c_size_t(S), C.c_size_t(nwords),
                                            C.c_int(int(mask_last)), _ptr(out)))

    def set_relin_key_slot(self, slot, ksk):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        self._chk(self.lib.hhe_set_relin_key_slot(self.h, C.c_int(slot), _ptr(ksk)))

    def fc_row(self, vi, w, W, n_inputs, out, B, relin_slot=0, default_galois_only=True, rk=None, gk=None):
        if rk is not None or gk is not None:  # the key objects the CSP names (CSP.cpp:306, 312-316)
            self._chk(self.lib.hhe_fc_row_ks(self.h, _ks(rk), _ks(gk), _ptr(vi), _ptr(w), C.c_size_t(W), C.c_size_t(n_inputs), _ptr(out), C.c_size_t(B)))
            return
        self._chk(self.lib.hhe_fc_row(self.h, _ptr(vi), _ptr(w), C.c_size_t(W), C.c_size_t(n_inputs), C.c_int(relin_slot),
                                      C.c_int(int(default_galois_only)), _ptr(out), C.c_size_t(B)))


    # ---- SEAL 4.0 wire format (parity unpinned, see include/hhe_gfx950.h) ----
    def seal_load_ciphertext(self, blob, out, offset=0):
        """blob: bytes-like; out: device uint64 buffer.  Returns (ct_size, parms_id bytes, consumed)."""
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
        size, used = C.c_size_t(0), C.c_size_t(0)
        pid = (C.c_uint8 * 32)()
        self._chk(self.lib.hhe_seal_load_ciphertext(self.h, C.byref(buf, offset), C.c_size_t(len(blob) - offset), _ptr(out),
                                                    C.c_size_t(out.numel() if hasattr(out, "numel") else out.size),
                                                    C.byref(size), pid, C.byref(used)))
        return int(size.value), bytes(pid), int(used.value)

    def seal_save_ciphertext(self, ct, ct_size, parms_id):
        need = C.c_size_t(0)
        pid = (C.c_uint8 * 32).from_buffer_copy(parms_id)
        words = ct_size * self.L * self.n
        cap = 16 + 32 + 1 + 40 + 16 + 8 + words * 8
        out = (C.c_uint8 * cap)()
        self._chk(self.lib.hhe_seal_save_ciphertext(self.h, _ptr(ct), C.c_size_t(ct_size), pid, out, C.c_size_t(cap), C.byref(need)))
        return bytes(out[:need.value])

    def seal_load_relin_keys(self, blob, slot=0):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
        used = C.c_size_t(0)
        self._chk(self.lib.hhe_seal_load_relin_keys(self.h, C.c_int(slot), buf, C.c_size_t(len(blob)), C.byref(used)))
        return int(used.value)

    def seal_load_galois_keys(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
        used, cnt = C.c_size_t(0), C.c_uint32(0)
        self._chk(self.lib.hhe_seal_load_galois_keys(self.h, buf, C.c_size_t(len(blob)), C.byref(used), C.byref(cnt)))
        return int(used.value), int(cnt.value)

    # ---- client / analyst ends ----
    def plain_keystream(self, key, first_block, nblocks, ks_out):
        """key: host uint64 [256]; ks_out device [nblocks][128]"""
        key = np.ascontiguousarray(key, dtype=np.uint64)
        assert key.shape == (2 * PASTA_T,)
        self._chk(self.lib.hhe_pasta3_plain_keystream(self.h, _ptr(key), C.c_uint64(first_block), C.c_size_t(nblocks), _ptr(ks_out)))

    def plain_crypt(self, key, records, S, nwords, out, decrypt=False):
        """records/out: device uint64 [S][nwords]"""
        key = np.ascontiguousarray(key, dtype=np.uint64)
        assert key.shape == (2 * PASTA_T,)
        self._chk(self.lib.hhe_pasta3_plain_crypt(self.h, _ptr(key), _ptr(records), C.c_size_t(S), C.c_size_t(nwords),
                                                  C.c_int(int(decrypt)), _ptr(out)))

    def decrypt(self, sk, ct, B, vals_out):
        """sk: host uint64 [K][N] (NTT form, key level); ct device [B][2][L][N]; vals_out device [B][N]"""
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        assert sk.size >= self.L * self.n
        self._chk(self.lib.hhe_decrypt(self.h, _ptr(sk), _ptr(ct), C.c_size_t(B), _ptr(vals_out)))


def bfv_default_coeff_modulus(n, lib=None):
    lib = lib or load_library()
    out = np.zeros(64, np.uint64)
    cnt = C.c_size_t(64)
    rc = lib.hhe_bfv_default_coeff_modulus(C.c_size_t(n), _ptr(out), C.byref(cnt))
    if rc:
        raise HheError(rc, lib.hhe_last_error().decode())
    return [int(v) for v in out[:cnt.value]]


def block_randomness(t, block_index, lib=None):
    lib = lib or load_library()
    mats = np.zeros((4, 2, PASTA_T, PASTA_T), np.uint64)
    rcs = np.zeros((4, 2, PASTA_T), np.uint64)
    rc = lib.hhe_pasta3_block_randomness(C.c_uint64(t), C.c_uint64(block_index), _ptr(mats), _ptr(rcs))
    if rc:
        raise HheError(rc, lib.hhe_last_error().decode())
    return mats, rcs
