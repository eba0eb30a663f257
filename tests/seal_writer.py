"""Test-side writer of SEAL 4.0 serialized objects (PARITY UNPINNED: a restatement of the layout pinned by the reference's
SEAL headers -- SEALHeader seal/serialization.h:60-93, DynArray::save_members seal/dynarray.h:652-680, KSwitchKeys members
seal/kswitchkeys.h:161-178, PublicKey = Ciphertext seal/publickey.h:89-93 -- and SEAL 4.0.0's published Ciphertext member
order; never compared with bytes SEAL itself produced, because the reference holds none and its libseal is never run)."""
import ctypes as C
import struct
import zlib

import numpy as np

MAGIC = 0xA15E
NONE, ZLIB, ZSTD = 0, 1, 2


def header(size, compr=NONE, major=4, minor=0):
    return struct.pack("<HBBBBHQ", MAGIC, 16, major, minor, compr, 0, size)


def obj(members, compr=NONE):
    """wrap already-serialized members into a SEAL object, compressing the members when asked"""
    if compr == ZLIB:
        members = zlib.compress(members)
    elif compr == ZSTD:
        members = zstd_compress(members)
    return header(16 + len(members), compr) + members


def zstd_compress(data):
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    cap = z.ZSTD_compressBound(len(data))
    out = C.create_string_buffer(cap)
    n = z.ZSTD_compress(out, cap, data, len(data), 3)
    assert not z.ZSTD_isError(n)
    return out.raw[:n]


def dynarray(words):
    words = np.ascontiguousarray(words, dtype="<u8").reshape(-1)
    return obj(struct.pack("<Q", words.size) + words.tobytes())


def ct_members(parms_id, words, size, n, cms, is_ntt=False, scale=1.0, corr=1):
    return (parms_id + struct.pack("<B", int(is_ntt)) + struct.pack("<QQQ", size, n, cms) + struct.pack("<d", scale) +
            struct.pack("<Q", corr) + dynarray(words))


def kswitch_keys(parms_id, table, n, K, compr=NONE):
    """table: list over the key index of None | array [L][2][K][N]"""
    m = parms_id + struct.pack("<Q", len(table))
    for entry in table:
        if entry is None:
            m += struct.pack("<Q", 0)
            continue
        m += struct.pack("<Q", entry.shape[0])
        for digit in entry:
            m += obj(ct_members(parms_id, digit, 2, n, K, is_ntt=True))
    return obj(m, compr)


def galois_table(gk):
    """GaloisKeys::get_index(elt) = (elt - 1) / 2 (seal/galoiskeys.h:48-74)"""
    table = [None] * max((int(e) - 1) // 2 + 1 for e in gk.elts)
    for e, k in zip(gk.elts, gk.keys):
        table[(int(e) - 1) // 2] = k
    return table


def parse_ciphertext(blob):
    """uncompressed Ciphertext object -> (parms_id, size, n, cms, words)"""
    magic, hs, major, _, compr, _, total = struct.unpack_from("<HBBBBHQ", blob)
    assert (magic, hs, major, compr, total) == (MAGIC, 16, 4, NONE, len(blob))
    pid = blob[16:48]
    is_ntt, size, n, cms, scale, corr = struct.unpack_from("<BQQQdQ", blob, 48)
    assert (is_ntt, scale, corr) == (0, 1.0, 1)
    at = 48 + 1 + 40
    _, _, _, _, icompr, _, isize = struct.unpack_from("<HBBBBHQ", blob, at)
    (count,) = struct.unpack_from("<Q", blob, at + 16)
    assert icompr == NONE and isize == 16 + 8 + count * 8 and at + isize == len(blob)
    return pid, size, n, cms, np.frombuffer(blob, dtype="<u8", count=count, offset=at + 24)
