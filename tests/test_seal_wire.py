"""SEAL 4.0 binary serialization at the boundary (csrc/hhe_seal_wire.cpp) on the tests-only emulator backend.

PARITY UNPINNED: the reference holds no serialized SEAL object and its prebuilt libseal is never run, so nothing here compares
against bytes SEAL produced.  The writer (tests/seal_writer.py) is a test-side restatement of the layout the reference's HEADERS pin
(SEALHeader seal/serialization.h:60-93, DynArray::save_members seal/dynarray.h:652-680, KSwitchKeys members
seal/kswitchkeys.h:161-178, PublicKey = Ciphertext seal/publickey.h:89-93) plus SEAL 4.0.0's published Ciphertext member
order; the checks are round trips, size arithmetic (Ciphertext::save_size) and behaviour after a load (a rotation with loaded
keys equals the oracle's)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from conftest import Setup
import parity_common as pc

from seal_writer import NONE, ZLIB, ZSTD, header, obj, ct_members, kswitch_keys, galois_table, parse_ciphertext

PID_DATA = bytes(range(32))
PID_KEY = bytes(range(100, 132))


@pytest.fixture(scope="module")
def mem():
    return pc.HostMem()


@pytest.fixture(scope="module")
def S(orc):
    return Setup(orc, 10, [50] * 4, extra_steps=(1, 3))


def fresh_cts(S, count, seed=0):
    rng = np.random.default_rng(seed)
    return [S.O.encrypt(S.pk, S.O.encode(rng.integers(0, S.t, S.n)), 40 + i) for i in range(count)]


def test_ciphertext_save_load_round_trip_and_save_size(api, emu_lib, mem, S):
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    ct = fresh_cts(S, 1)[0]
    blob = X.seal_save_ciphertext(mem.to_dev(ct), 2, PID_DATA)
    # Ciphertext::save_size(none): header + parms_id + is_ntt + 3 sizes + scale + correction factor + DynArray(header + count + data)
    assert len(blob) == 16 + 32 + 1 + 24 + 8 + 8 + (16 + 8 + ct.size * 8)
    assert blob == obj(ct_members(PID_DATA, ct, 2, S.n, X.L))
    pid, size, n, cms, words = parse_ciphertext(blob)
    assert (pid, size, n, cms) == (PID_DATA, 2, S.n, X.L) and (words.reshape(ct.shape) == ct).all()
    out = mem.empty(S.O.ct_shape)
    size, pid, used = X.seal_load_ciphertext(blob, out)
    assert (size, pid, used) == (2, PID_DATA, len(blob))
    assert (mem.to_host(out) == ct).all()
    # a size-3 product travels too (the analyst never sees one, the spill file can)
    X2 = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    ct3 = S.O.multiply(ct, ct)
    blob3 = X2.seal_save_ciphertext(mem.to_dev(ct3), 3, PID_DATA)
    out3 = mem.empty(ct3.shape)
    assert X2.seal_load_ciphertext(blob3, out3)[0] == 3
    assert (mem.to_host(out3) == ct3).all()


@pytest.mark.parametrize("compr", [ZLIB, ZSTD])
def test_compressed_ciphertexts_inflate_to_the_same_words(api, emu_lib, mem, S, compr):
    # SEAL's default compr_mode is zstd when built with it, zlib otherwise (seal/serialization.h compr_mode_default)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    ct = fresh_cts(S, 1, seed=compr)[0]
    blob = obj(ct_members(PID_DATA, ct, 2, S.n, X.L), compr)
    assert blob[5] == compr
    out = mem.empty(S.O.ct_shape)
    size, _, used = X.seal_load_ciphertext(blob, out)
    assert size == 2 and used == len(blob)
    assert (mem.to_host(out) == ct).all()


def test_concatenated_stream_and_spill_file(api, emu_lib, mem, S):
    """CSP.cpp:552-605 sends ciphertexts back to back in one byte string; :495-547 spills `size_t count` + that stream"""
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    cts = fresh_cts(S, 3, seed=9)
    modes = [NONE, ZSTD, ZLIB]
    stream = b"".join(obj(ct_members(PID_DATA, c, 2, S.n, X.L), m) for c, m in zip(cts, modes))
    spill = struct.pack("<Q", len(cts)) + stream
    (count,) = struct.unpack_from("<Q", spill)
    at = 8
    for i in range(count):
        out = mem.empty(S.O.ct_shape)
        _, _, used = X.seal_load_ciphertext(spill, out, offset=at)
        assert (mem.to_host(out) == cts[i]).all()
        at += used
    assert at == len(spill)


@pytest.mark.parametrize("compr", [NONE, ZSTD])
def test_galois_and_relin_keys_loaded_from_blobs_drive_the_ops(api, emu_lib, mem, S, compr):
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    gblob = kswitch_keys(PID_KEY, galois_table(S.gk), S.n, X.K, compr)
    used, cnt = X.seal_load_galois_keys(gblob)
    assert used == len(gblob) and cnt == len(S.gk.elts)
    assert all(X.has_galois_key(int(e)) for e in S.gk.elts)
    rblob = kswitch_keys(PID_KEY, [S.rk], S.n, X.K, compr)  # RelinKeys::get_index(2) = 0
    assert X.seal_load_relin_keys(rblob) == len(rblob)
    cts = np.stack(fresh_cts(S, 2, seed=4))
    d = mem.to_dev(cts)
    out = mem.empty(cts.shape)
    for step in (1, 3, -1):
        X.rotate_rows(d, step, out, 2)
        assert (mem.to_host(out)[1] == O.rotate_rows(cts[1], step, S.gk)[0]).all(), step
    X.rotate_columns(d, out, 2)
    assert (mem.to_host(out)[0] == O.rotate_columns(cts[0], S.gk)).all()
    o3 = mem.empty((2, 3) + cts.shape[2:])
    X.multiply(d, d, o3, 2)
    X.relinearize(o3, out, 2)
    assert (mem.to_host(out)[1] == O.relinearize(O.multiply(cts[1], cts[1]), S.rk)).all()


def test_malformed_and_foreign_objects_are_rejected(api, emu_lib, mem, S, orc):
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    ct = fresh_cts(S, 1, seed=2)[0]
    good = obj(ct_members(PID_DATA, ct, 2, S.n, X.L))
    out = mem.empty(S.O.ct_shape)

    def bad(blob, what):
        with pytest.raises(RuntimeError, match=what):
            X.seal_load_ciphertext(blob, out)

    bad(b"\x00\x00" + good[2:], "bad magic")
    bad(good[:10], "shorter than a SEALHeader")
    bad(good[:-8], "truncated")
    bad(good[:5] + b"\x07" + good[6:], "unknown compression mode")
    bad(good[:3] + b"\x02" + good[4:], "unsupported SEAL major version")
    bad(obj(ct_members(PID_DATA, ct, 2, S.n, X.L, is_ntt=True)), "NTT form")
    bad(obj(ct_members(PID_DATA, ct, 2, S.n, X.L, scale=2.0)), "scale / correction factor")
    bad(obj(ct_members(PID_DATA, ct[:, :-1], 2, S.n, X.L - 1)), "not at the data level")
    bad(obj(ct_members(PID_DATA, ct, 2, S.n // 2, X.L * 2)), "ciphertext size fields out of range|not at the data level")
    # a seeded ciphertext stores c1 as a seed: fewer words than size*L*N
    bad(obj(ct_members(PID_DATA, ct.reshape(-1)[:ct.size // 2 + 9], 2, S.n, X.L)), "seeded")
    over = ct.copy()
    over[1, 0, 5] = S.q[0]
    bad(obj(ct_members(PID_DATA, over, 2, S.n, X.L)), "not reduced")
    bad(header(16 + 100, ZLIB) + b"\x01" * 100, "zlib data is corrupt")
    bad(header(16 + 100, ZSTD) + b"\x01" * 100, "zstd data is corrupt")
    z = zlib.compress(good[16:])
    bad(header(16 + len(z) - 20, ZLIB) + z[:-20], "zlib data is truncated")
    small_out = mem.empty((S.n,))
    with pytest.raises(RuntimeError, match="output buffer too small"):
        X.seal_load_ciphertext(good, small_out)
    # keys of another parameter set / wrong shape
    with pytest.raises(RuntimeError, match="digits"):
        X.seal_load_galois_keys(kswitch_keys(PID_KEY, [S.rk[:-1]], S.n, X.K))
    with pytest.raises(RuntimeError, match="not a size-2, key-level, NTT-form"):
        X.seal_load_relin_keys(obj(PID_KEY + struct.pack("<QQ", 1, X.L) +
                                   b"".join(obj(ct_members(PID_KEY, d[:, :-1], 2, S.n, X.K - 1, is_ntt=True)) for d in S.rk)))
    with pytest.raises(RuntimeError, match="beyond key\\(2\\)"):
        X.seal_load_relin_keys(kswitch_keys(PID_KEY, [S.rk, S.rk], S.n, X.K))
    with pytest.raises(RuntimeError, match="holds no key"):
        X.seal_load_relin_keys(kswitch_keys(PID_KEY, [None], S.n, X.K))
    # save: capacity is reported, not overrun
    need = C.c_size_t(0)
    pid = (C.c_uint8 * 32).from_buffer_copy(PID_DATA)
    tiny = (C.c_uint8 * 8)()
    rc = emu_lib.hhe_seal_save_ciphertext(X.h, pc_ptr(mem.to_dev(ct)), C.c_size_t(2), pid, tiny, C.c_size_t(8), C.byref(need))
    assert rc == 6 and need.value == len(good)  # HHE_ERR_CAPACITY


def pc_ptr(a):
    return C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("compr", [ZLIB, ZSTD])
def test_compression_bombs_are_refused_before_they_are_inflated(api, emu_lib, mem, S, compr):
    """the blobs arrive over gRPC: a few KB of zlib / zstd must not inflate into gigabytes -- the decoder stops at the largest
    legitimate object of the context and reports an error (never a std::bad_alloc across the C boundary)"""
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    out = mem.empty(S.O.ct_shape)
    bomb = obj(b"\x00" * (64 << 20), compr)          # 64 MiB of zeros: a few dozen KB compressed
    assert len(bomb) < (1 << 20)
    with pytest.raises(RuntimeError, match="inflates beyond the largest object"):
        X.seal_load_ciphertext(bomb, out)
    # key objects are bounded as well (at most 64 keys of this context's size: ~1.4 MB each here)
    big = obj(b"\x00" * (200 << 20), compr)
    with pytest.raises(RuntimeError, match="inflates beyond the largest object"):
        X.seal_load_galois_keys(big)
    with pytest.raises(RuntimeError, match="inflates beyond the largest object"):
        X.keyset().seal_load_relin_keys(big)
    # a compressed NESTED array (never written by SEAL) is refused before anything is inflated
    ct = fresh_cts(S, 1)[0]
    words = np.ascontiguousarray(ct, dtype="<u8").reshape(-1)
    nested = obj(struct.pack("<Q", words.size) + words.tobytes(), compr)
    members = PID_DATA + struct.pack("<B", 0) + struct.pack("<QQQ", 2, S.n, X.L) + struct.pack("<d", 1.0) + struct.pack("<Q", 1) + nested
    with pytest.raises(RuntimeError, match="nested object is compressed"):
        X.seal_load_ciphertext(obj(members), out)


def test_key_loads_are_all_or_nothing(api, emu_lib, mem, S):
    """SEAL's load(context, ..) swaps the object in only on success: a GaloisKeys blob whose LAST key is bad (truncated table, foreign
    shape, unreduced word) leaves the target set exactly as it was -- no earlier key of the blob has been uploaded"""
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    ks = X.keyset()
    table = galois_table(S.gk)
    good = kswitch_keys(PID_KEY, table, S.n, X.K)
    idx = [i for i, e in enumerate(table) if e is not None]
    bad_word = [None if e is None else e.copy() for e in table]
    bad_word[idx[-1]][0, 0, 0, 3] = S.q[0]                 # not reduced
    for blob, what in ((good[:len(good) - 4096], "truncated|exceeds the buffer"),
                       (kswitch_keys(PID_KEY, bad_word, S.n, X.K), "not reduced"),
                       (kswitch_keys(PID_KEY, table[:-1] + [S.rk[:-1]], S.n, X.K), "digits")):
        if "truncated" in what:   # keep the header's size field consistent with the shortened buffer, so the table itself is short
            blob = header(len(blob)) + blob[16:]
        with pytest.raises(RuntimeError, match=what):
            ks.seal_load_galois_keys(blob)
        assert not any(ks.has_galois(int(e)) for e in S.gk.elts)
        with pytest.raises(RuntimeError, match=what):
            X.seal_load_galois_keys(blob)
        assert not any(X.has_galois_key(int(e)) for e in S.gk.elts)
    # and the good blob loads into a SET, which then drives rotations with exactly these keys
    used, cnt = ks.seal_load_galois_keys(good)
    assert used == len(good) and cnt == len(S.gk.elts) and not X.has_galois_key(int(S.gk.elts[0]))
    ct = fresh_cts(S, 1, seed=5)[0]
    out = mem.empty((1,) + O.ct_shape)
    X.rotate_rows(mem.to_dev(ct[None]), 3, out, 1, gk=ks)
    assert (mem.to_host(out)[0] == O.rotate_rows(ct, 3, S.gk)[0]).all()
    with pytest.raises(RuntimeError, match="Galois key not present"):
        X.rotate_rows(mem.to_dev(ct[None]), 3, out, 1)   # the default set is still empty
    # SEAL 4.0.0's member order: scale (the double 1.0) THEN correction factor (the integer 1); the swapped order is refused
    m = PID_DATA + struct.pack("<B", 0) + struct.pack("<QQQ", 2, S.n, X.L) + struct.pack("<Q", 1) + struct.pack("<d", 1.0)
    words = np.ascontiguousarray(ct, dtype="<u8").reshape(-1)
    swapped = obj(m + obj(struct.pack("<Q", words.size) + words.tobytes()))
    with pytest.raises(RuntimeError, match="scale / correction factor"):
        X.seal_load_ciphertext(swapped, mem.empty(O.ct_shape))
