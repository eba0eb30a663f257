"""GPU parity: libhhe_gfx950.so (hand-written gfx950 kernels, through the C ABI) against the CPU oracle,
bit for bit, on identical seeded inputs.  Run on an MI355X: python -m pytest tests -m gpu."""
import numpy as np
import pytest

from conftest import Setup
import parity_common as pc

pytestmark = pytest.mark.gpu
T = 65537


@pytest.fixture(scope="module")
def mem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return pc.TorchMem("cuda:0")


@pytest.fixture(scope="module")
def lib(api):
    lib = api.load_library()  # fails loudly if the HIP library is missing
    assert lib.hhe_backend() == b"hip-gfx950"
    return lib


@pytest.mark.parametrize("logn,bits", [(10, [50] * 3), (11, [60] * 3), (12, [55] * 2), (13, [60] * 2), (14, [50] * 2), (15, [60] * 4)])
def test_ntt_all_sizes(orc, api, lib, mem, logn, bits):
    q = orc.coeff_modulus_create(1 << logn, bits)
    O = orc.Oracle(logn, q, T)
    X = api.Context(logn, q, T, lib=lib)
    pc.check_context_constants(X, O)
    pc.check_ntt(X, O, mem, seed=logn)


def test_ntt_n65536(orc, api, lib, mem):
    t = 8088322049
    q = orc.coeff_modulus_create(1 << 16, [60] * 3)
    pc.check_ntt(api.Context(16, q, t, lib=lib), orc.Oracle(16, q, t), mem, seed=16)


def test_every_op_bit_exact_small(orc, api, lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    pc.check_ops(X, small, mem, B=3)


def test_every_op_bit_exact_n4096_l5(orc, api, lib, mem):
    S = Setup(orc, 12, [45] * 6)
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    pc.check_ops(X, S, mem, B=2, seed=9)


def test_transcipher_ragged_blocks(orc, api, lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    pt = [(7 * i + 3) % 256 for i in range(300)]
    pc.check_transcipher(X, small, orc, mem, pt)


def test_transcipher_is_deterministic_and_cache_independent(orc, api, lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    pt = [(5 * i + 1) % 256 for i in range(256)]
    a = pc.check_transcipher(X, small, orc, mem, pt, oracle_items=[0])
    X.clear_block_cache()
    b = pc.check_transcipher(X, small, orc, mem, pt, oracle_items=[])
    assert (a == b).all()


def test_config2_n32768_4primes_batch(orc, api, lib, mem):
    """BASELINE config 2 shape (N=2^15, 4x60-bit primes, t=65537, block counter 0).  Noise budget is 0 with
    this modulus, so parity is on ciphertext words: item 0 against the oracle, and every other
    item through linearity out_s - out_0 == scaled(encode(c_s) - encode(c_0)) (the keystream ciphertext is
    common to the batch)."""
    S = Setup(orc, 15, [60] * 4)
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    B = 8
    cw = np.zeros((B, 128), np.uint64)
    for s in range(B):
        x = np.array([(7 * i + 3 + s) % 256 for i in range(128)], dtype=np.uint64)
        cw[s] = orc.pasta_encrypt(S.t, S.key, x)
    out = mem.empty((B,) + O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, [128] * B, [0] * B, out)
    res = mem.to_host(out)
    assert (res[0] == O.transcipher_block(S.enc_key, S.rk, S.gk, cw[0], 0)).all()
    zero = np.zeros(O.ct_shape, np.uint64)
    base = O.sub_plain(res[0], O.encode(cw[0]))  # = -KS
    for s in range(1, B):
        assert (res[s] == O.add_plain(base, O.encode(cw[s]))).all()
    assert zero.sum() == 0


def test_fc_row_and_flatten_small(orc, api, lib, mem):
    """packed FC row (multiply + relinearize + NAF rotation sum) and flatten, N=1024 with all default Galois keys."""
    S = Setup(orc, 10, [50] * 9, all_galois=True)
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    rng = np.random.default_rng(4)
    n_in = 37
    v = rng.integers(0, 4, n_in)
    w = rng.integers(-8, 9, n_in)
    vi = O.encrypt(S.pk, O.encode(v), 21)
    wc = O.encrypt(S.pk, O.encode(w), 22)
    out = mem.empty((1,) + O.ct_shape)
    X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, out, 1)
    ref, ks = O.fc_row(vi, wc, S.rk, S.gk, n_in)
    got = mem.to_host(out)[0]
    assert (got == ref).all()
    # the reference's own check: FC result == plain integer matmul (hhe_pktnn_examples.cpp:692-699)
    dec = O.decode(O.decrypt(S.sk, got))
    assert int(dec[n_in - 1]) == int(np.dot(v, w)) % S.t
    # flatten of 3 blocks
    blocks = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, 256, 128)), 30 + i) for i in range(3)])
    fo = mem.empty((1,) + O.ct_shape)
    X.flatten(mem.to_dev(blocks[None]), 3, fo, 1)
    assert (mem.to_host(fo)[0] == O.flatten(blocks, S.gk)).all()


BFV_DEFAULT_16384 = [281474976546817, 281474976317441, 281474975662081, 562949952798721, 562949952700417,
                     562949952274433, 562949951979521, 562949951881217, 562949951619073]


@pytest.fixture(scope="module")
def cfg1(orc):
    """the reference's default parameters (t=65537, N=16384, BFVDefault 9 primes; configs/config.cpp:19-20) with every key the
    MNIST flow needs: all default Galois elements (Analyst.cpp:62-65) plus the PASTA / flatten steps"""
    S = Setup.__new__(Setup)
    S.t, S.logn, S.n, S.q = T, 14, 1 << 14, BFV_DEFAULT_16384
    S.O = O = orc.Oracle(14, S.q, T)
    S.sk = O.keygen_secret(1)
    S.pk = O.keygen_public(S.sk, 2)
    S.rk = O.keygen_relin(S.sk, 3)
    steps = [-1, 0, 128] + [-128 * i for i in range(1, 7)]
    elts = list(dict.fromkeys([int(e) for e in O.galois_elts_all()] + [O.galois_elt(s) for s in steps]))
    S.gk = O.keygen_galois(S.sk, elts, 7)
    S.key = np.array([(i * 2654435761 + 12345) % T for i in range(256)], dtype=np.uint64)
    S.enc_key = O.encrypt(S.pk, O.pasta_pack_key(S.key), 11)
    return S


def test_config1_dry_run_mnist_sample_end_to_end(orc, api, lib, mem, cfg1):
    """BASELINE config 1 shape (reference defaults: t=65537, N=16384, BFVDefault 9 primes): one 784-word 2-bit
    sample -> 7 transcipherings -> mask (last block) -> flatten -> one FC row (multiply + relinearize +
    encrypted_vec_sum over all default Galois keys), checked the way the reference checks itself
    (hhe_pktnn_examples.cpp:639-648, 692-699): decrypt == input, FC == plain integer matmul."""
    S, O = cfg1, cfg1.O
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mnist_1fc.json")))
    n_in = 784
    pix = np.array(fx["pixels"][0], dtype=np.int64)          # MNIST test image 0, 2-bit quantised (config 1 / 3 input)
    W = np.array(fx["weights_rows"], dtype=np.int64)         # the reference's 1-layer integer weights, one row per neuron
    w = W[0]
    cw, ncw = S.sym_blocks(orc, pix)
    nb = cw.shape[0]
    assert nb == 7 and ncw[-1] == 16
    blocks = mem.empty((nb,) + O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, list(range(nb)), blocks)
    hb = mem.to_host(blocks)
    # ragged last block, bit for bit against the oracle
    assert (hb[6] == O.transcipher_block(S.enc_key, S.rk, S.gk, cw[6, :16], 6)).all()
    # mask the last block (rem = 16), flatten, decrypt == input
    last = mem.to_dev(hb[6:7])
    X.mask(last, np.ones(16, np.uint64), last, 1)
    hb2 = hb.copy()
    hb2[6] = mem.to_host(last)[0]
    flat = mem.empty((1,) + O.ct_shape)
    X.flatten(mem.to_dev(hb2[None]), nb, flat, 1)
    dec = O.decode(O.decrypt(S.sk, mem.to_host(flat)[0]))
    assert (dec[:n_in] == pix).all()
    # FC row
    wc = O.encrypt(S.pk, O.encode(w % T), 22)
    out = mem.empty((1,) + O.ct_shape)
    X.fc_row(flat, mem.to_dev(wc[None]), 1, n_in, out, 1)
    res = O.decode(O.decrypt(S.sk, mem.to_host(out)[0]))
    assert int(res[n_in - 1]) == int(np.dot(pix, w)) % T
    assert O.noise_budget(S.sk, mem.to_host(out)[0], 8) > 0
    # the whole protocol on the device for the real samples (all fixture images): client PASTA encryption, CSP decompose, all
    # 10 FC rows in one call, analyst decryption -- logits equal the plain integer layer, prediction equals the label
    wcs = np.stack([O.encrypt(S.pk, O.encode(W[r] % T), 40 + r) for r in range(10)])
    for img in range(len(fx["pixels"])):
        pix_i = np.array(fx["pixels"][img], dtype=np.int64)
        d_sym = mem.empty((1, n_in))
        X.plain_crypt(S.key, mem.to_dev(pix_i.astype(np.uint64)[None]), 1, n_in, d_sym)
        assert (mem.to_host(d_sym)[0] == orc.pasta_encrypt(T, S.key, pix_i)).all()
        flat2 = mem.empty((1,) + O.ct_shape)
        X.decompose(mem.to_dev(S.enc_key), mem.to_host(d_sym), flat2, mask_last=True)
        if img == 0:
            assert (mem.to_host(flat2) == mem.to_host(flat)).all()
        vi10 = mem.to_dev(np.repeat(mem.to_host(flat2), 10, axis=0))
        out10 = mem.empty((10,) + O.ct_shape)
        X.fc_row(vi10, mem.to_dev(wcs), 10, n_in, out10, 10)
        vals = mem.empty((10, O.n))
        X.decrypt(S.sk, out10, 10, vals)
        got = mem.to_host(vals)[:, n_in - 1].astype(np.int64)
        logits = np.where(got > (T + 1) // 2, got - T, got)
        assert [int(v) for v in logits] == fx["plain_logits"][img]
        assert int(np.argmax(logits)) == fx["labels"][img] == fx["argmax"][img]


def test_config5_n65536_six_primes_rotation_chain_and_multiply(orc, api, lib, mem):
    """BASELINE config 5 shape: N=2^16, 6x60-bit primes, t=8088322049; key-switch heavy chain, ciphertext parity."""
    t = 8088322049
    S = Setup(orc, 16, [60] * 6, t=t)
    O = S.O
    X = api.Context(S.logn, S.q, t, lib=lib)
    S.load_keys(X)
    rng = np.random.default_rng(5)
    B = 2
    cts = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, 1 << 30, O.n)), 40 + b) for b in range(B)])
    d = mem.to_dev(cts)
    refs = [c.copy() for c in cts]
    for _ in range(4):
        X.rotate_rows(d, -1, d, B)
        refs = [O.rotate_rows(r, -1, S.gk)[0] for r in refs]
    h = mem.to_host(d)
    for b in range(B):
        assert (h[b] == refs[b]).all()
    o3 = mem.empty((B, 3, O.L, O.n))
    X.multiply(d, d, o3, B)
    assert (mem.to_host(o3)[0] == O.multiply(refs[0], refs[0])).all()
    out = mem.empty((B,) + O.ct_shape)
    X.relinearize(o3, out, B)
    assert (mem.to_host(out)[0] == O.relinearize(mem.to_host(o3)[0], S.rk)).all()
    # one whole transciphering at these parameters (the noise budget is exhausted here, SURVEY 3.4: ciphertext words are
    # compared, not decryptions): 518 key switches through the fused pipeline at N = 2^16, L = 5, item 0 against the oracle
    key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
    pt = np.array([(7 * i + 3) % 256 for i in range(128)], dtype=np.uint64)
    cw = orc.pasta_encrypt(t, key, pt).reshape(1, 128)
    enc_key = O.encrypt(S.pk, O.pasta_pack_key(key), 11)
    tr = mem.empty((2,) + O.ct_shape)
    X.transcipher(mem.to_dev(enc_key), np.concatenate([cw, cw]), [128, 128], [0, 0], tr)
    h = mem.to_host(tr)
    assert (h[0] == O.transcipher_block(enc_key, S.rk, S.gk, cw[0], 0)).all() and (h[1] == h[0]).all()


def test_config5_rotation_chain_128_steps(orc, api, lib, mem):
    """BASELINE config 5's chain at depth: 128 steps of rotate_rows(-1) at N = 2^16, 6 x 60-bit primes on two ciphertexts -- every step's
    rounding feeds the next, so any deviation in the generic key switch (digit loads through the Galois map, fused row kernel with both
    sums inverse-transformed, mod-down in the store of the last inverse pass) shows in the words.  Compared with the SHA-256 of the
    oracle's words after 4, 64 and 128 steps (tests/golden/config5_chain.json, computed once on the CPU: make_config5_chain.py)."""
    import hashlib
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_chain.json")))
    t = 8088322049
    S = Setup(orc, 16, [60] * 6, t=t)
    O = S.O
    X = api.Context(S.logn, S.q, t, lib=lib)
    S.load_keys(X)
    rng = np.random.default_rng(5)
    cts = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, 1 << 30, O.n)), 40 + b) for b in range(2)])
    src, dst = mem.to_dev(cts), mem.empty(cts.shape)
    for step in range(1, 129):
        X.rotate_rows(src, -1, dst, 2)
        src, dst = dst, src
        if str(step) in fx["steps"]:
            h = mem.to_host(src)
            assert [hashlib.sha256(np.ascontiguousarray(h[b]).tobytes()).hexdigest() for b in range(2)] == fx["steps"][str(step)], step


def test_babystep_giantstep_variant(orc, api, lib, mem):
    S = Setup(orc, 10, [50] * 9, extra_steps=[-16 * k for k in range(1, 8)])
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    pt = np.array([(13 * i + 2) % 256 for i in range(140)], dtype=np.uint64)
    cw, ncw = S.sym_blocks(orc, pt)
    out = mem.empty((2,) + S.O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, [0, 1], out, use_bsgs=True)
    res = mem.to_host(out)
    for b in range(2):
        assert (res[b] == S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b, use_bsgs=True)).all()
        assert (S.O.decode(S.O.decrypt(S.sk, res[b]))[:ncw[b]] == pt[b * 128:b * 128 + ncw[b]]).all()


@pytest.mark.parametrize("knobs", [
    {"HHE_STREAMS": "0"}, {"HHE_STREAMS": "2"}, {"HHE_STREAMS": "3", "HHE_CHUNK": "1"}, {"HHE_MATMUL": "0", "HHE_STREAMS": "1"},
])
def test_execution_knobs_are_result_neutral(orc, api, lib, mem, small, monkeypatch, knobs):
    pt = [(3 * i + 1) % 256 for i in range(300)]
    cw, ncw = small.sym_blocks(orc, pt)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    out = mem.empty((3,) + small.O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, ncw, [0, 1, 2], out)
    res = mem.to_host(out)
    for b in (0, 2):
        assert (res[b] == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[b, :ncw[b]], b)).all(), knobs


def test_config2_full_batch_256(orc, api, lib, mem):
    """BASELINE config 2 at its full size (256 blocks per GPU: 2 chunks over 2 internal streams): item 0 and
    item 255 against the oracle, every other item through the size-independent linearity property, and a checksum of
    checksums across the batch."""
    S = Setup(orc, 15, [60] * 4)
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    B = 256
    rng = np.random.default_rng(2)
    cw = rng.integers(0, S.t, size=(B, 128), dtype=np.uint64)
    out = mem.empty((B,) + O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, [128] * B, [0] * B, out)
    res = mem.to_host(out)
    for i in (0, B - 1):
        assert (res[i] == O.transcipher_block(S.enc_key, S.rk, S.gk, cw[i], 0)).all()
    base = O.sub_plain(res[0], O.encode(cw[0]))  # = -KS, common to the batch (block counter 0)
    for i in range(1, B - 1, 17):
        assert (res[i] == O.add_plain(base, O.encode(cw[i]))).all()
    # all items differ only in c0 (add_plain touches c0 only): c1 must be identical across the batch
    assert (res[:, 1] == res[0, 1]).all()


def test_decompose_on_device(orc, api, lib, mem):
    """BaseCSP::decompose glue (blocks -> mask -> flatten) for two records on the GPU"""
    S = Setup(orc, 10, [50] * 9, extra_steps=(-128, -256))
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    pts = [np.array([(7 * i + 3 + s) % 256 for i in range(300)], dtype=np.uint64) for s in range(2)]
    recs = np.stack([orc.pasta_encrypt(S.t, S.key, p) for p in pts])
    out = mem.empty((2,) + O.ct_shape)
    X.decompose(mem.to_dev(S.enc_key), recs, out, mask_last=True)
    res = mem.to_host(out)
    for s in range(2):
        cw, ncw = S.sym_blocks(orc, pts[s])
        blocks = [O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b) for b in range(3)]
        blocks[2] = O.mask(blocks[2], np.ones(44, np.uint64))
        assert (res[s] == O.flatten(np.stack(blocks), S.gk)).all()
        assert (O.decode(O.decrypt(S.sk, res[s]))[:300] == pts[s]).all()


@pytest.mark.parametrize("t,logn,bits", [(65537, 10, [50, 50]), (8088322049, 12, [55, 55]), (1096486890805657601, 10, [60, 60])])
def test_client_plain_pasta_matches_reference_built_golden(orc, api, lib, mem, t, logn, bits):
    """SURVEY 8f-4 on the GPU: PASTA::encrypt / decrypt / keystream kernels vs vectors from the reference's pasta_3_plain.cpp"""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pasta_plain.json")))
    X = api.Context(logn, orc.coeff_modulus_create(1 << logn, bits), t, lib=lib)
    pc.check_plain_cipher_golden(X, orc, mem, g)


def test_client_bulk_keystream_many_counters(orc, api, lib, mem, small):
    """one launch over 1000 block counters: spot checks against the oracle + all blocks distinct"""
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    key = pc.golden_key(small.t)
    nb = 1000
    ks = mem.empty((nb, 128))
    X.plain_keystream(key, 7, nb, ks)
    got = mem.to_host(ks)
    for b in (0, 1, 63, 64, 511, 999):
        assert (got[b] == orc.pasta_keystream(small.t, key, 7 + b)).all()
    assert len({row.tobytes() for row in got}) == nb
    assert int(got.max()) < small.t


def test_analyst_batched_decrypt(orc, api, lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    pc.check_decrypt(X, small, mem, B=5)


def test_end_to_end_client_encrypt_csp_transcipher_analyst_decrypt(orc, api, lib, mem):
    """the whole protocol on the device: PASTA-encrypt (client) -> decompose (CSP) -> decrypt (analyst) returns the input"""
    S = Setup(orc, 10, [50] * 9, extra_steps=(-128, -256))
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    rng = np.random.default_rng(11)
    pts = rng.integers(0, 4, (2, 300), dtype=np.uint64)
    d_ct = mem.empty((2, 300))
    X.plain_crypt(S.key, mem.to_dev(pts), 2, 300, d_ct)
    recs = mem.to_host(d_ct)
    for s in range(2):
        assert (recs[s] == orc.pasta_encrypt(S.t, S.key, pts[s])).all()
    out = mem.empty((2,) + S.O.ct_shape)
    X.decompose(mem.to_dev(S.enc_key), recs, out, mask_last=True)
    vals = mem.empty((2, S.O.n))
    X.decrypt(S.sk, out, 2, vals)
    got = mem.to_host(vals)
    assert (got[:, :300] == pts).all() and not got[:, 300:S.O.n // 2].any()


def test_fc_row_shared_digit_variants(orc, api, lib, mem, monkeypatch):
    S = Setup(orc, 11, [60] * 4, all_galois=True)
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=lib), S, orc, mem, monkeypatch, n_in=100)


def test_fc_row_variants_bench_shape_n32768(orc, api, lib, mem, monkeypatch):
    """the FC's execution variants at the bench's parameters (N = 2^15, 4 x 60-bit primes: L = 3, K = 4): ks_perm_row_kernel<8> with the
    LDS twiddle heap, leaf groups across trie nodes, the slot pool -- every word against the oracle; then the same without the fused
    row kernel of the non-leaf children (HHE_FC_ROWFUSED=0)"""
    S = Setup(orc, 15, [60] * 4, all_galois=True)
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=lib), S, orc, mem, monkeypatch, n_in=14)
    monkeypatch.setenv("HHE_FC_ROWFUSED", "0")
    pc.check_fc_variants(lambda: api.Context(S.logn, S.q, S.t, lib=lib), S, orc, mem, monkeypatch, n_in=14)


def test_config4_two_layer_chain(orc, api, lib, mem):
    """BASELINE config 4: the reference's ECG first-layer weights (weights/ecg/ecg_512/fc1_weight_50epochs_bz4.csv, fixture
    tests/golden/ecg_fc1.json) on seeded synthetic 128-word inputs in [0,255] (the reference's ECG inputs are missing)."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ecg_fc1.json")))
    S = Setup(orc, 12, [55] * 5, all_galois=True)
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    x = np.random.default_rng(44).integers(0, 256, 128)
    pc.check_two_layer_chain(X, S, mem, n_in=128, w1_vals=fx["fc1_weight"], x_vals=x)


def test_transcipher_unaffected_by_interleaved_eager_work(orc, api, lib, mem, small):
    """CSP flow: transcipher, then hundreds of other launches (FC, packed ops), then transcipher again -- identical words.
    (Round 1 found hipGraph replays of the matmul loop going wrong after ~200 eager launches on other streams; the graph
    path was removed in round 2 after the defect persisted with HHE_MERGE=0, see DESIGN.md.)"""
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    small.load_keys(X)
    pt = [(5 * i + 2) % 256 for i in range(300)]
    cw, ncw = small.sym_blocks(orc, pt)
    nb = len(ncw)
    first = mem.empty((nb,) + small.O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, ncw, list(range(nb)), first)
    ref = mem.to_host(first).copy()
    d = mem.to_dev(ref[:1])
    o = mem.empty((1,) + small.O.ct_shape)
    for _ in range(400):
        X.add(d, d, o, 1)
    again = mem.empty((nb,) + small.O.ct_shape)
    X.transcipher(mem.to_dev(small.enc_key), cw, ncw, list(range(nb)), again)
    assert (mem.to_host(again) == ref).all()
    assert (ref[0] == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[0, :ncw[0]], 0)).all()


def test_seal_streams_in_transcipher_stream_out(orc, api, lib, mem, small):
    """the CSP's data movement for one request (CSP.cpp:328-490 keys and the HE-encrypted PASTA key arrive as SEAL streams,
    :552-605 results leave as one): every object crosses the boundary as bytes.  Layout parity unpinned (tests/seal_writer.py);
    the arithmetic in between is bit-exact against the oracle as everywhere else."""
    import seal_writer as sw
    pid_d, pid_k = bytes(range(32)), bytes(range(64, 96))
    X = api.Context(small.logn, small.q, small.t, lib=lib)
    X.seal_load_relin_keys(sw.kswitch_keys(pid_k, [small.rk], small.n, X.K, sw.ZSTD))
    _, cnt = X.seal_load_galois_keys(sw.kswitch_keys(pid_k, sw.galois_table(small.gk), small.n, X.K, sw.ZSTD))
    assert cnt == len(small.gk.elts)
    enc_key = mem.empty(small.O.ct_shape)
    size, pid, _ = X.seal_load_ciphertext(sw.obj(sw.ct_members(pid_d, small.enc_key, 2, small.n, X.L), sw.ZLIB), enc_key)
    assert size == 2 and (mem.to_host(enc_key) == small.enc_key).all()
    pt = [(11 * i + 5) % 256 for i in range(200)]
    cw, ncw = small.sym_blocks(orc, pt)
    out = mem.empty((2,) + small.O.ct_shape)
    X.transcipher(enc_key, cw, ncw, np.arange(2), out)
    stream = b"".join(X.seal_save_ciphertext(out[b], 2, pid) for b in range(2))
    half = len(stream) // 2
    for b in range(2):
        rpid, rsize, rn, rcms, words = sw.parse_ciphertext(stream[b * half:(b + 1) * half])
        assert (rpid, rsize, rn, rcms) == (pid_d, 2, small.n, X.L)
        assert (words.reshape(small.O.ct_shape) == small.O.transcipher_block(small.enc_key, small.rk, small.gk, cw[b, :ncw[b]], b)).all()


def test_config3_mnist_64_images_whole_protocol(orc, api, lib, mem, cfg1):
    """BASELINE config 3 at fixture scale: the first 64 MNIST test images (tests/golden/mnist_64.json, 2-bit pixels) through the
    device protocol in ONE batch -- client PASTA encryption, CSP decompose (448 transcipherings + mask + flatten), the 784 x 10
    FC (640 rows in one call, the CSP's own RelinKeys object for the relinearization and the analyst's default GaloisKeys object
    for the slot sums, as CSP.cpp:306, 312-316 names them), analyst decryption.  The reference's acceptance test
    (hhe_pktnn_examples.cpp:692-699, 861-862): HHE logits == plain integer matmul for every image, prediction == label
    wherever the plain model is right."""
    import json
    import os
    S, O = cfg1, cfg1.O
    here = os.path.dirname(os.path.abspath(__file__))
    fx = json.load(open(os.path.join(here, "golden", "mnist_64.json")))
    W = np.array(json.load(open(os.path.join(here, "golden", "mnist_1fc.json")))["weights_rows"], dtype=np.int64)
    packed = np.array([list(bytes.fromhex(h)) for h in fx["pixels_2bit_hex"]], dtype=np.uint8)
    pix = ((packed[:, :, None] >> (2 * np.arange(4))) & 3).reshape(len(packed), 784).astype(np.int64)
    NS, n_in = pix.shape
    assert NS == 64 and (pix @ W.T == np.array(fx["plain_logits"])).all()
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    pasta, analyst_gk, csp_gk, csp_rk = X.keyset(), X.keyset(), X.keyset(), X.keyset()
    pasta.set_relin(S.rk)
    dflt = {int(e) for e in O.galois_elts_all()}
    for e, k in zip(S.gk.elts, S.gk.keys):
        e = int(e)
        if e in dflt:
            analyst_gk.set_galois(e, k)
        pasta.set_galois(e, k)
    # csp_gk = add_gk_indices + the flatten steps -128 i of a 7-block record (hhe_pktnn_examples.cpp:601-615), its own randomness
    gk_flat = O.keygen_galois(S.sk, [int(O.galois_elt(s)) for s in [0, -1, 128] + [-128 * i for i in range(1, 7)]], 71)
    for e, k in zip(gk_flat.elts, gk_flat.keys):
        csp_gk.set_galois(int(e), k)
    rk_csp = O.keygen_relin(S.sk, 72)
    csp_rk.set_relin(rk_csp)
    d_sym = mem.empty((NS, n_in))
    X.plain_crypt(S.key, mem.to_dev(pix.astype(np.uint64)), NS, n_in, d_sym)
    recs = mem.to_host(d_sym)
    assert (recs[5] == orc.pasta_encrypt(T, S.key, pix[5])).all()
    flat = mem.empty((NS,) + O.ct_shape)
    X.decompose(mem.to_dev(S.enc_key), recs, flat, mask_last=True, rk=pasta, gk=pasta, flatten_gk=csp_gk)
    # one record against the oracle's op sequence with the same three key objects
    cw, ncw = S.sym_blocks(orc, pix[63])
    blocks = [O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b) for b in range(7)]
    blocks[6] = O.mask(blocks[6], np.ones(16, np.uint64))
    assert (mem.to_host(flat)[63] == O.flatten(np.stack(blocks), gk_flat)).all()
    wcs = np.stack([O.encrypt(S.pk, O.encode(W[r] % T), 40 + r) for r in range(10)])
    import torch
    vi = flat.repeat_interleave(10, dim=0).contiguous()   # item = (sample, neuron), neuron = item % 10
    out = mem.empty((NS * 10,) + O.ct_shape)
    X.fc_row(vi, mem.to_dev(wcs), 10, n_in, out, NS * 10, rk=csp_rk, gk=analyst_gk)
    vals = mem.empty((NS * 10, O.n))
    X.decrypt(S.sk, out, NS * 10, vals)
    got = mem.to_host(vals)[:, n_in - 1].astype(np.int64).reshape(NS, 10)
    logits = np.where(got > (T + 1) // 2, got - T, got)
    plain = np.array(fx["plain_logits"])
    assert ((logits - plain) % T == 0).all()
    pred = logits.argmax(axis=1)
    right = np.array(fx["argmax"]) == np.array(fx["labels"])
    assert right.sum() == 58 and (pred[right] == np.array(fx["labels"])[right]).all() and (pred == np.array(fx["argmax"])).all()
    del torch


def test_key_sets_have_identity(orc, api, lib, mem):
    """two GaloisKeys objects and two RelinKeys objects under one secret key on the device (CSP.cpp:238-242, 271-278, 306, 312-316):
    N = 4096 runs rotations / relinearize through the fused row kernel and its per-key Shoup tables, N = 2048 the separate kernels"""
    for logn, bits in ((12, [55] * 4), (11, [50] * 4)):
        S = Setup(orc, logn, bits)
        X = api.Context(S.logn, S.q, S.t, lib=lib)   # default set empty
        pc.check_key_sets(X, S, orc, mem)
        X.close()


@pytest.mark.parametrize("pattern", ["max", "alt", "max_keys"])
def test_matmul_loop_adversarial_residues(orc, api, lib, mem, pattern):
    """worst-case residues through the fused matmul loop at 60-bit primes: N = 4096 (L = 2) and the metric's N = 2^15 (L = 3)"""
    for logn in (12, 15):
        S = Setup(orc, logn, [60] * (3 if logn == 12 else 4))
        X = api.Context(S.logn, S.q, S.t, lib=lib)
        pc.check_matmul_adversarial(X, S, orc, mem, pattern)
        X.close()
