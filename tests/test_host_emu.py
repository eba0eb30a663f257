"""CPU checks of the PRODUCT's host driver (csrc/hhe_api.cpp, hhe_context.cpp, hhe_pasta_public.cpp) and
of the kernel bodies' index arithmetic (csrc/hhe_kernel_bodies.h) through the tests-only emulator
backend (tests/emu).  The emulator is test infrastructure; the product library has no CPU path."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import Setup
import parity_common as pc

HERE = os.path.dirname(os.path.abspath(__file__))
T = 65537


@pytest.fixture(scope="module")
def mem():
    return pc.HostMem()


@pytest.mark.parametrize("logn,bits", [(10, [50] * 3), (11, [60] * 3), (12, [55] * 2), (13, [60] * 2), (14, [50] * 2), (15, [60] * 2)])
def test_ntt_all_pass_sizes(orc, api, emu_lib, mem, logn, bits):
    q = orc.coeff_modulus_create(1 << logn, bits)
    O = orc.Oracle(logn, q, T)
    X = api.Context(logn, q, T, lib=emu_lib)
    assert b"emulator" in emu_lib.hhe_backend()
    pc.check_context_constants(X, O)
    pc.check_ntt(X, O, mem, seed=logn)


def test_ntt_n65536_plain_modulus_33bit(orc, api, emu_lib, mem):
    t = 8088322049  # configs/config.cpp:22 ; 65537 cannot batch at N=2^16
    q = orc.coeff_modulus_create(1 << 16, [60, 60])
    O = orc.Oracle(16, q, t)
    X = api.Context(16, q, t, lib=emu_lib)
    pc.check_ntt(X, O, mem, seed=16)


def test_every_op_bit_exact(orc, api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    pc.check_ops(X, small, mem, B=3)


def test_ops_l8_exercises_lazy_accumulator_flush(orc, api, emu_lib, mem):
    # L=5 > 4 limbs: the 128-bit lazy sums are flushed mid-way (ks_mac / BEHZ conversions)
    S = Setup(orc, 10, [40] * 6)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    pc.check_ops(X, S, mem, B=2, seed=5)


def test_transcipher_ragged_blocks_bit_exact_and_decrypts(orc, api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    pt = [(7 * i + 3) % 256 for i in range(300)]  # 3 blocks, last one ragged (44 words)
    pc.check_transcipher(X, small, orc, mem, pt)


def test_transcipher_same_block_counter_batch(orc, api, emu_lib, mem, small):
    # BASELINE config 2 shape: independent 128-word inputs, all with block counter 0
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    O = small.O
    B = 2
    cw = np.zeros((B, 128), np.uint64)
    for s in range(B):
        x = np.array([(7 * i + 3 + s) % 256 for i in range(128)], dtype=np.uint64)
        cw[s] = orc.pasta_encrypt(small.t, small.key, x)[:128]
    out = mem.empty((B,) + O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, [128] * B, [0] * B, out)
    res = mem.to_host(out)
    for s in range(B):
        assert (res[s] == O.transcipher_block(small.enc_key, small.rk, small.gk, cw[s], 0)).all()
        assert (O.decode(O.decrypt(small.sk, res[s]))[:128] == [(7 * i + 3 + s) % 256 for i in range(128)]).all()


def test_missing_keys_and_bad_args_fail_loudly(orc, api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    O = small.O
    cw = np.zeros((1, 128), np.uint64)
    out = mem.empty((1,) + O.ct_shape)
    with pytest.raises(api.HheError) as e:
        X.transcipher(mem.to_dev(small.enc_key), cw, [128], [0], out)
    assert e.value.code == api.ERR_NO_RELIN_KEY
    X.set_relin_key(small.rk)
    with pytest.raises(api.HheError) as e:
        X.transcipher(mem.to_dev(small.enc_key), cw, [128], [0], out)
    assert e.value.code == api.ERR_NO_GALOIS_KEY and "Galois key not present" in str(e.value)
    with pytest.raises(api.HheError):
        X.rotate_rows(mem.to_dev(small.enc_key[None]), 5, out, 1)  # no key for 5 = naf{1,4}
    with pytest.raises(api.HheError):
        api.Context(10, [small.q[0], 12345], small.t, lib=emu_lib)  # not an NTT prime
    with pytest.raises(api.HheError):
        api.Context(10, small.q, 65539, lib=emu_lib)  # t not 1 mod 2N


def test_product_block_randomness_matches_reference_golden(api, emu_lib):
    g = json.load(open(os.path.join(HERE, "golden", "pasta_plain.json")))
    for c in g["randomness"]:
        mats, rcs = api.block_randomness(c["t"], c["block"], lib=emu_lib)
        assert hashlib.sha256(mats.tobytes()).hexdigest() == c["mats_sha256"]
        assert hashlib.sha256(rcs.tobytes()).hexdigest() == c["rcs_sha256"]


def test_fused_matmul_equals_op_by_op_schedule(orc, api, emu_lib, mem, small, monkeypatch):
    """the 20-transform fused diagonal pipeline and the literal op-by-op schedule give the same words"""
    pt = [(11 * i + 5) % 256 for i in range(128)]
    cw, ncw = small.sym_blocks(orc, pt)
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("HHE_MATMUL", mode)
        X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
        small.load_keys(X)
        out = mem.empty((1,) + small.O.ct_shape)
        X.transcipher(mem.to_dev(small.enc_key), cw, ncw, [5], out)
        outs.append(mem.to_host(out))
    assert (outs[0] == outs[1]).all()
    assert (outs[0][0] == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[0], 5)).all()


def test_mixed_prime_sizes_force_digit_reduction(orc, api, emu_lib, mem):
    # 50-bit data primes vs 36-bit ones: q_I >= 4 q_J, so key-switch digits must be reduced before NTT_J
    S = Setup(orc, 10, [36, 36, 50, 50, 50])
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    pc.check_ops(X, S, mem, B=2, seed=11)


def test_babystep_giantstep_variant_bit_exact(orc, api, emu_lib, mem):
    """PASTA_SEAL::babystep_giantstep (pasta_3_seal.cpp:267-366; activate_bsgs) -- different ciphertext words than the
    diagonal method, same plaintext; compared with the oracle's restatement and decrypted."""
    S = Setup(orc, 10, [50] * 9, extra_steps=[-16 * k for k in range(1, 8)])
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    pt = np.array([(13 * i + 2) % 256 for i in range(140)], dtype=np.uint64)
    cw, ncw = S.sym_blocks(orc, pt)
    out = mem.empty((2,) + S.O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, [0, 1], out, use_bsgs=True)
    res = mem.to_host(out)
    for b in range(2):
        ref = S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b, use_bsgs=True)
        assert (res[b] == ref).all()
        assert (S.O.decode(S.O.decrypt(S.sk, res[b]))[:ncw[b]] == pt[b * 128:b * 128 + ncw[b]]).all()
    plain = S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[0], 0, use_bsgs=False)
    assert not (res[0] == plain).all()  # a different (reference-defined) ciphertext
    Y = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    Y.set_relin_key(S.rk)
    for i, e in enumerate(S.gk.elts[:3]):
        Y.set_galois_key(int(e), S.gk.keys[i])
    with pytest.raises(api.HheError):
        Y.transcipher(mem.to_dev(S.enc_key), cw, ncw, [0, 1], out, use_bsgs=True)  # giant-step keys missing


def test_fc_row_naf_trie_equals_sequential_rotations(orc, api, emu_lib, mem):
    """hhe_fc_row evaluates the NAF rotation chains of encrypted_vec_sum as a prefix trie; the words must equal the
    oracle's literal loop (sealhelper.cpp:379-392), including the ±N/2 skip rule (n > N/4) and key policies."""
    S = Setup(orc, 10, [50] * 9, all_galois=True, extra_steps=(-300,))
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    X.set_relin_key_slot(1, S.rk)
    rng = np.random.default_rng(4)
    for n_in in (37, 400):
        v = rng.integers(0, 4, n_in)
        w = rng.integers(-8, 9, n_in)
        vi = O.encrypt(S.pk, O.encode(v), 21)
        wc = O.encrypt(S.pk, O.encode(w), 22)
        out = mem.empty((1,) + O.ct_shape)
        # the context also holds a key for step -300: with default_galois_only the call must ignore it ...
        X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, out, 1, relin_slot=1, default_galois_only=True)
        dflt = orc.GaloisKeys(S.gk.elts[:-1], S.gk.keys[:-1])
        ref, ks = O.fc_row(vi, wc, S.rk, dflt, n_in)
        got = mem.to_host(out)[0]
        assert (got == ref).all()
        assert int(O.decode(O.decrypt(S.sk, got))[n_in - 1]) == int(np.dot(v, w)) % S.t
        # ... and with the union policy it must use it, exactly like SEAL's has_key() shortcut
        X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, out, 1, relin_slot=0, default_galois_only=False)
        ref2, _ = O.fc_row(vi, wc, S.rk, S.gk, n_in)
        assert (mem.to_host(out)[0] == ref2).all()
        assert n_in <= 300 or not (ref == ref2).all()


def test_fc_row_leaf_data_limbs_through_c1_sums(orc, api, emu_lib, mem, monkeypatch):
    """the data-limb sums of the leaf key switches come from per-Galois-element INTEGER sums of the parents' c1 (one inner product per
    element; 64-bit words + a byte that counts their wraps): with 400 inputs and 60-bit primes ~100 leaves share an element, the sums wrap
    2^64 several times, and the words are the oracle's -- as they are with one inner product per leaf (HHE_FC_CSUM=0)"""
    S = Setup(orc, 10, [60] * 4, all_galois=True)
    O = S.O
    rng = np.random.default_rng(9)
    n_in, B = 400, 2
    v, w = rng.integers(0, 4, n_in), rng.integers(-8, 9, n_in)
    wc = O.encrypt(S.pk, O.encode(w), 52)
    vi = np.stack([O.encrypt(S.pk, O.encode(v), 50 + b) for b in range(B)])
    refs = [O.fc_row(vi[b], wc, S.rk, S.gk, n_in)[0] for b in range(B)]
    for knob in ("1", "0"):
        monkeypatch.setenv("HHE_FC_CSUM", knob)
        X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
        S.load_keys(X)
        out = mem.empty((B,) + O.ct_shape)
        X.fc_row(mem.to_dev(vi), mem.to_dev(wc[None]), 1, n_in, out, B, relin_slot=0, default_galois_only=False)
        got = mem.to_host(out)
        for b in range(B):
            assert (got[b] == refs[b]).all(), (knob, b)
        closes = X.query("fc_csum_closes")
        assert (0 < closes <= 2 * 8) if knob == "1" else closes == 0   # one close per distinct last-term element and chunk
        assert X.query("fc_fallbacks") == 0
        X.close()


def test_decompose_record_mask_flatten(orc, api, emu_lib, mem):
    """BaseCSP::decompose on device: blocks -> mask(last) -> flatten for a batch of records, vs the oracle's op sequence"""
    S = Setup(orc, 10, [50] * 9, extra_steps=(-128, -256))
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    nwords, nrec = 300, 2
    recs, pts = [], []
    for s in range(nrec):
        pt = np.array([(7 * i + 3 + s) % 256 for i in range(nwords)], dtype=np.uint64)
        pts.append(pt)
        recs.append(orc.pasta_encrypt(S.t, S.key, pt))
    out = mem.empty((nrec,) + O.ct_shape)
    X.decompose(mem.to_dev(S.enc_key), np.stack(recs), out, mask_last=True)
    res = mem.to_host(out)
    for s in range(nrec):
        cw, ncw = S.sym_blocks(orc, pts[s])
        blocks = [O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b) for b in range(3)]
        blocks[2] = O.mask(blocks[2], np.ones(44, np.uint64))
        ref = O.flatten(np.stack(blocks), S.gk)
        assert (res[s] == ref).all()
        assert (O.decode(O.decrypt(S.sk, res[s]))[:nwords] == pts[s]).all()


def test_reference_n65536_29_prime_context_is_accepted(orc, api, emu_lib, mem):
    """the reference's hard-coded N=65536 parameter set (SEAL_Cipher.cpp:50-60: 29 primes, L=28): context tables and
    a 28-limb ciphertext add / NTT round trip (the deep BEHZ / key-switch arrays are sized for it)"""
    t = 8088322049
    q = api.bfv_default_coeff_modulus(65536, emu_lib)
    X = api.Context(16, q, t, lib=emu_lib)
    assert X.L == 28
    assert X.query("root", 0) == orc.minimal_primitive_root(1 << 17, q[0])
    rng = np.random.default_rng(0)
    a = np.stack([rng.integers(0, q[j], 1 << 16, dtype=np.uint64) for j in range(28)])
    d = mem.to_dev(a)
    X.ntt(d, 28, 0, 28, False)
    X.ntt(d, 28, 0, 28, True)
    assert (mem.to_host(d) == a).all()


@pytest.mark.parametrize("knobs", [
    {"HHE_STREAMS": "0"}, {"HHE_STREAMS": "2"}, {"HHE_STREAMS": "3", "HHE_CHUNK": "1"}, {"HHE_MATMUL": "0", "HHE_STREAMS": "1"},
])
def test_every_execution_knob_gives_the_same_words(orc, api, emu_lib, mem, small, monkeypatch, knobs):
    """chunking / streams / the literal op-by-op schedule only change scheduling"""
    pt = [(3 * i + 1) % 256 for i in range(300)]
    cw, ncw = small.sym_blocks(orc, pt)
    refs = [small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[b, :ncw[b]], b) for b in range(3)]
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    out = mem.empty((3,) + small.O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, ncw, [0, 1, 2], out)
    res = mem.to_host(out)
    for b in range(3):
        assert (res[b] == refs[b]).all(), knobs


def test_edge_cases_single_word_block_and_large_block_counter(orc, api, emu_lib, mem, small):
    """ragged extremes: a 1-word block, a full block and a far-away block counter in one batch; empty batch is rejected"""
    O = small.O
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    counters = [0, (1 << 32) + 5]
    cw = np.zeros((2, 128), np.uint64)
    pts = []
    for i, ctr in enumerate(counters):
        nw = 1 if i == 0 else 128
        pt = np.array([(9 * j + 4 + i) % 256 for j in range(nw)], dtype=np.uint64)
        ks = orc.pasta_keystream(small.t, small.key, ctr)
        cw[i, :nw] = (pt + ks[:nw]) % small.t
        pts.append(pt)
    out = mem.empty((2,) + O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, [1, 128], counters, out)
    res = mem.to_host(out)
    for i, ctr in enumerate(counters):
        nw = len(pts[i])
        assert (res[i] == O.transcipher_block(small.enc_key, small.rk, small.gk, cw[i, :nw], ctr)).all()
        assert (O.decode(O.decrypt(small.sk, res[i]))[:nw] == pts[i]).all()
    with pytest.raises(api.HheError):
        X.transcipher(mem.to_dev(small.enc_key), np.zeros((0, 128), np.uint64), [], [], out)
    with pytest.raises(api.HheError):
        X.transcipher(mem.to_dev(small.enc_key), cw, [129, 128], counters, out)  # more than 128 words in a block


@pytest.mark.parametrize("t,logn,bits", [(65537, 10, [50, 50]), (8088322049, 10, [55, 55]), (1096486890805657601, 10, [60, 60])])
def test_client_plain_pasta_matches_reference_built_golden(orc, api, emu_lib, mem, t, logn, bits):
    """device PASTA::encrypt/decrypt/keystream vs vectors produced by the reference's own pasta_3_plain.cpp."""
    g = json.load(open(os.path.join(HERE, "golden", "pasta_plain.json")))
    X = api.Context(logn, orc.coeff_modulus_create(1 << logn, bits), t, lib=emu_lib)
    pc.check_plain_cipher_golden(X, orc, mem, g)


def test_client_plain_pasta_rejects_key_words_above_modulus(api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    key = np.full(256, small.t, np.uint64)
    with pytest.raises(api.HheError) as e:
        X.plain_keystream(key, 0, 1, mem.empty((1, 128)))
    assert e.value.code == api.ERR_INVALID


def test_analyst_batched_decrypt_matches_oracle(orc, api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    pc.check_decrypt(X, small, mem)


def test_fc_row_shared_digit_variants(orc, api, emu_lib, mem, monkeypatch):
    S = Setup(orc, 10, [50] * 4, all_galois=True)
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=emu_lib), S, orc, mem, monkeypatch)


def test_config4_two_layer_chain(orc, api, emu_lib, mem):
    S = Setup(orc, 10, [50] * 4, all_galois=True)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    pc.check_two_layer_chain(X, S, mem)


@pytest.mark.parametrize("logn,bits", [(12, [50, 50, 50]), (13, [40] * 6)])
def test_fused_row_kernel_full_tiles(orc, api, emu_lib, mem, logn, bits):
    """N >= 4096: the matmul loop runs ks_row_kernel (row pass of the digit transforms + key inner product with Shoup key
    products + inverse row pass in one kernel; the c0 branch rides in its grid).  Same words as the oracle; L = 5 also
    exercises the fold of the lazy sums after the fourth digit."""
    S = Setup(orc, logn, bits)
    pt = [(13 * i + 7) % 256 for i in range(100)]
    cw, ncw = S.sym_blocks(orc, pt)
    ref = S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[0, :ncw[0]], 3)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    out = mem.empty((2,) + S.O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), np.concatenate([cw, cw]), [ncw[0], ncw[0]], [3, 3], out)
    res = mem.to_host(out)
    assert (res[0] == ref).all() and (res[1] == ref).all()
    # the literal op-by-op schedule (HHE_MATMUL=0 at context creation) gives the same words
    os.environ["HHE_MATMUL"] = "0"
    try:
        X0 = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    finally:
        del os.environ["HHE_MATMUL"]
    S.load_keys(X0)
    X0.transcipher(mem.to_dev(S.enc_key), cw, [ncw[0]], [3], out)
    assert (mem.to_host(out)[0] == ref).all()


def test_row_kernel_lds_twiddle_heap_n32768(orc, api, emu_lib, mem):
    """N = 2^15 (256-point rows): ks_row_kernel stages the twiddles of its first two rounds in LDS, for the digit
    transforms, the inverse transforms and the c0 tiles of its grid.  One whole transciphering (matmul loop: c0 tiles in the
    grid) and the generic key switches (rotation, relinearize: S_0 inverse-transformed too), same words as the oracle."""
    S = Setup(orc, 15, [50, 50, 50])
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    O = S.O
    rng = np.random.default_rng(15)
    ct = O.encrypt(S.pk, O.encode(rng.integers(0, S.t, S.n)), 9)
    d = mem.to_dev(ct[None])
    out = mem.empty((1,) + O.ct_shape)
    X.rotate_rows(d, -1, out, 1)
    assert (mem.to_host(out)[0] == O.rotate_rows(ct, -1, S.gk)[0]).all()
    o3 = mem.empty((1, 3) + O.ct_shape[1:])
    X.multiply(d, d, o3, 1)
    X.relinearize(o3, out, 1)
    assert (mem.to_host(out)[0] == O.relinearize(O.multiply(ct, ct), S.rk)).all()
    pt = [(5 * i + 1) % 256 for i in range(128)]
    cw, ncw = S.sym_blocks(orc, pt)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, [0], out)
    assert (mem.to_host(out)[0] == O.transcipher_block(S.enc_key, S.rk, S.gk, cw[0], 0)).all()


def test_two_threads_on_one_context_are_serialised(orc, api, emu_lib, mem, small):
    """the reference's gRPC handlers call one cipher object concurrently (CSPRPC.cpp:201-203): every C-ABI entry point
    holds the context's lock, so two threads on ONE context get the same words as sequential calls"""
    import threading
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    pts = [[(3 * i + 1 + 17 * k) % 256 for i in range(128)] for k in range(2)]
    blocks = [small.sym_blocks(orc, pt) for pt in pts]
    refs = [small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[0], 4 + k) for k, (cw, _) in enumerate(blocks)]
    outs = [mem.empty((1,) + small.O.ct_shape) for _ in range(2)]
    errs = []

    def work(k):
        try:
            cw, ncw = blocks[k]
            for _ in range(2):
                X.transcipher(mem.to_dev(small.enc_key), cw, ncw, [4 + k], outs[k])
                d = mem.to_dev(mem.to_host(outs[k]))
                X.add(d, d, d, 1)  # generic ops interleave with the other thread's transciphering
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert (mem.to_host(outs[k])[0] == refs[k]).all()


def test_generic_key_switch_through_the_row_kernel(orc, api, emu_lib, mem):
    """N >= 4096: Evaluator::switch_key (rotations by any key, relinearize) also runs through ks_row_kernel, here with the
    inverse row pass of S_0 as well and the mod-down fused into the store of the last inverse pass (STORE_KSF)"""
    S = Setup(orc, 12, [50, 50, 50, 50])
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    S.load_keys(X)
    pc.check_ops(X, S, mem, B=2, seed=21)


def test_fc_row_variants_full_tiles(orc, api, emu_lib, mem, monkeypatch):
    """the FC's execution variants at N = 4096 (full tiles): the epilogues that carry the mod-down (STORE_KSF with the
    Galois-gathered base) and the leaf sums (STORE_RACC with q_sp * galois(c0)) on the tile geometry the GPU runs"""
    S = Setup(orc, 12, [50] * 3, all_galois=True)
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=emu_lib), S, orc, mem, monkeypatch, n_in=21)
    # non-leaf children without ks_perm_row_kernel (inner product and inverse row passes as separate launches)
    monkeypatch.setenv("HHE_FC_ROWFUSED", "0")
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=emu_lib), S, orc, mem, monkeypatch, n_in=21)


@pytest.mark.parametrize("pattern", ["max", "alt", "max_keys"])
def test_matmul_loop_adversarial_residues(orc, api, emu_lib, mem, pattern):
    """worst-case residues through the fused matmul loop at 60-bit primes (N = 4096: the row kernel, its truncated Shoup
    products and pseudo-Mersenne folds); the emulator is built with -DHHE_RANGE_CHECK, so a lazy sum that wraps 64 bits or a
    lazy difference that goes negative aborts instead of hiding behind a congruent result"""
    S = Setup(orc, 12, [60, 60, 60])
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    pc.check_matmul_adversarial(X, S, orc, mem, pattern)


def test_block_table_cache_is_bounded_lru(orc, api, emu_lib, mem, small):
    """the per-block public tables are cached per block counter; the cache holds at most its byte limit: least recently used counters
    go first, never one the running call uses, and an evicted counter is rebuilt with the same words"""
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    small.load_keys(X)
    pt = [(3 * i + 1) % 256 for i in range(128)]
    out = mem.empty((1,) + small.O.ct_shape)

    def run(ctr):
        ks = orc.pasta_keystream(small.t, small.key, ctr)
        cw = ((np.array(pt, dtype=np.uint64) + ks) % small.t).reshape(1, 128)
        X.transcipher(mem.to_dev(small.enc_key), cw, [128], [ctr], out)
        return mem.to_host(out)[0].copy(), cw

    first, cw0 = run(0)
    per = X.query("block_cache_bytes")
    assert per > 0 and X.query("block_cache_entries") == 1
    X.set_block_cache_limit(2 * per)
    run(1)
    run(2)                       # counter 0 is the least recently used: dropped
    assert X.query("block_cache_entries") == 2 and X.query("block_cache_bytes") == 2 * per
    run(1)                       # refresh 1; then 3 evicts 2
    run(3)
    assert X.query("block_cache_entries") == 2
    again, _ = run(0)            # rebuilt: identical words
    assert (again == first).all() and (first == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw0[0], 0)).all()
    # one call that needs more counters than the limit holds is served (its own tables are pinned while it runs)
    cws = np.concatenate([run(c)[1] for c in (4, 5, 6)])
    out3 = mem.empty((3,) + small.O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cws, [128] * 3, [4, 5, 6], out3)
    assert X.query("block_cache_entries") == 3
    assert (mem.to_host(out3)[2] == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cws[2], 6)).all()
    X.set_block_cache_limit(0)
    assert X.query("block_cache_entries") == 0 and X.query("block_cache_bytes") == 0
