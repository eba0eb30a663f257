"""Parity checks shared by the CPU-emulator suite (numpy memory) and the GPU suite
(torch cuda memory).  Every check compares the C-ABI library with the oracle bit for bit."""
import numpy as np


class HostMem:
    """numpy-backed 'device' memory for the tests-only emulator."""

    def to_dev(self, a):
        return np.ascontiguousarray(a, dtype=np.uint64).copy()

    def empty(self, shape):
        return np.zeros(shape, np.uint64)

    def to_host(self, b):
        return np.array(b, copy=True)


class TorchMem:
    """torch cuda tensors (int64 storage viewed as uint64 words)."""

    def __init__(self, device="cuda:0"):
        import torch
        self.torch, self.device = torch, device

    def to_dev(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return self.torch.from_numpy(a.view(np.int64)).to(self.device)

    def empty(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.int64, device=self.device)

    def to_host(self, b):
        self.torch.cuda.synchronize()
        return b.cpu().numpy().view(np.uint64)


def check_context_constants(X, O):
    for i in range(O.K):
        assert X.query("root", i) == O.query("root", i)
    for i in range(O.L + 1):
        assert X.query("bsk", i) == O.query("bsk", i)
    assert X.query("gamma") == O.query("gamma")
    for s in (-1, 1, 128, -128, 0):
        assert X.query("galois_elt", s) == O.galois_elt(s)
    for j in range(O.L):
        assert X.query("delta", j) == O.query("delta", j)


def check_ntt(X, O, mem, seed=0):
    rng = np.random.default_rng(seed)
    nm = 2 * O.K  # coeff primes + Bsk primes
    polys = np.stack([rng.integers(0, 1 << 40, O.n, dtype=np.uint64) for _ in range(nm + 1)])
    polys[nm] %= O.t
    ref = np.stack([O.ntt_fwd(i, polys[i]) for i in range(nm)] + [O.ntt_fwd(-1, polys[nm])])
    d = mem.to_dev(polys)
    X.ntt(d, nm + 1, 0, nm + 1, False)
    assert (mem.to_host(d) == ref).all()
    X.ntt(d, nm + 1, 0, nm + 1, True)
    assert (mem.to_host(d) == polys).all()
    # extreme residues: the lazily reduced butterflies (values allowed to grow to 16q between folds) must not wrap 64 bits
    mods = [O.q[i] if i < O.K else O.query("bsk", i - O.K) for i in range(nm)] + [O.t]
    for pattern in ("max", "alt", "one_hot"):
        ext = np.zeros((nm + 1, O.n), np.uint64)
        for i, qi in enumerate(mods):
            if pattern == "max":
                ext[i, :] = qi - 1
            elif pattern == "alt":
                ext[i, ::2] = qi - 1
            else:
                ext[i, O.n - 1] = qi - 1
        ref = np.stack([O.ntt_fwd(i, ext[i]) for i in range(nm)] + [O.ntt_fwd(-1, ext[nm])])
        d = mem.to_dev(ext)
        X.ntt(d, nm + 1, 0, nm + 1, False)
        assert (mem.to_host(d) == ref).all(), pattern
        X.ntt(d, nm + 1, 0, nm + 1, True)
        assert (mem.to_host(d) == ext).all(), pattern


def check_ops(X, S, mem, B=3, seed=0):
    """S: conftest.Setup with keys loaded into X."""
    O = S.O
    n, t = O.n, O.t
    rng = np.random.default_rng(seed)
    cts = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, t, n)), 10 + b) for b in range(B)])
    d_cts = mem.to_dev(cts)
    out = mem.empty(cts.shape)
    # encode (ragged count)
    vals = np.stack([rng.integers(0, t, 100, dtype=np.uint64) for _ in range(B)])
    d_pl = mem.empty((B, n))
    X.encode(mem.to_dev(vals), B, 100, d_pl)
    pl = mem.to_host(d_pl)
    for b in range(B):
        assert (pl[b] == O.encode(vals[b])).all()
    # add / negate
    rev = cts[::-1].copy()
    X.add(d_cts, mem.to_dev(rev), out, B)
    h = mem.to_host(out)
    for b in range(B):
        assert (h[b] == O.add(cts[b], rev[b])).all()
    X.negate(d_cts, out, B)
    assert (mem.to_host(out)[0] == O.negate(cts[0])).all()
    # add_plain / sub_plain / broadcast
    X.add_plain(d_cts, d_pl, out, B)
    h = mem.to_host(out)
    for b in range(B):
        assert (h[b] == O.add_plain(cts[b], pl[b])).all()
    X.add_plain(d_cts, d_pl, out, B, subtract=True)
    assert (mem.to_host(out)[1] == O.sub_plain(cts[1], pl[1])).all()
    X.add_plain(d_cts, mem.to_dev(pl[0:1]), out, B, bcast=True)
    assert (mem.to_host(out)[B - 1] == O.add_plain(cts[B - 1], pl[0])).all()
    # multiply_plain
    X.multiply_plain(d_cts, d_pl, out, B)
    h = mem.to_host(out)
    for b in range(B):
        assert (h[b] == O.multiply_plain(cts[b], pl[b])).all()
    X.multiply_plain(d_cts, mem.to_dev(pl[1:2]), out, B, bcast=True)
    assert (mem.to_host(out)[B - 1] == O.multiply_plain(cts[B - 1], pl[1])).all()
    # galois / rotations (incl. in place)
    for e, k in zip(S.gk.elts, S.gk.keys):
        X.apply_galois(d_cts, int(e), out, B)
        h = mem.to_host(out)
        for b in range(B):
            assert (h[b] == O.apply_galois(cts[b], int(e), k)).all(), int(e)
    x = mem.to_dev(cts)
    X.rotate_rows(x, -1, x, B)
    assert (mem.to_host(x)[1] == O.rotate_rows(cts[1], -1, S.gk)[0]).all()
    X.rotate_columns(d_cts, out, B)
    assert (mem.to_host(out)[0] == O.rotate_columns(cts[0], S.gk)).all()
    # BEHZ multiply / square / relinearize
    o3 = mem.empty((B, 3, O.L, n))
    X.multiply(d_cts, mem.to_dev(rev), o3, B)
    h3 = mem.to_host(o3)
    for b in range(B):
        assert (h3[b] == O.multiply(cts[b], rev[b])).all()
    X.multiply(d_cts, d_cts, o3, B)
    h3 = mem.to_host(o3)
    assert (h3[1] == O.multiply(cts[1], cts[1])).all()
    X.relinearize(o3, out, B)
    assert (mem.to_host(out)[1] == O.relinearize(h3[1], S.rk)).all()


def check_transcipher(X, S, orc, mem, pt, block_ids=None, check_decrypt=True, oracle_items=None):
    """transcipher the PASTA encryption of pt; compare chosen items with the oracle bit for bit."""
    O = S.O
    cw, ncw = S.sym_blocks(orc, pt)
    nb = cw.shape[0]
    block_ids = list(range(nb)) if block_ids is None else block_ids
    out = mem.empty((nb,) + O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, block_ids, out)
    res = mem.to_host(out)
    for b in (range(nb) if oracle_items is None else oracle_items):
        ref = O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], block_ids[b])
        assert (res[b] == ref).all(), f"block {b} differs from oracle"
        if check_decrypt:
            dec = O.decode(O.decrypt(S.sk, res[b]))[:ncw[b]]
            assert (dec == np.asarray(pt[b * 128:b * 128 + ncw[b]], dtype=np.uint64)).all()
    return res


def golden_key(t):
    return np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)


def check_plain_cipher_golden(X, orc, mem, golden):
    """Client-side plain PASTA-3 on the device vs the vectors the reference's own pasta_3_plain.cpp produced
    (tests/golden/pasta_plain.json) and vs the oracle for block counters the fixture does not hold."""
    t = X.t
    key = golden_key(t)
    hit = 0
    for c in golden["randomness"]:
        if c["t"] != t:
            continue
        ks = mem.empty((1, 128))
        X.plain_keystream(key, c["block"], 1, ks)
        assert [int(v) for v in mem.to_host(ks)[0]] == c["keystream"]
        hit += 1
    assert hit, "no golden keystream for this modulus"
    # a run of consecutive counters in one launch, each block against the oracle
    nb = 5
    ks = mem.empty((nb, 128))
    X.plain_keystream(key, 3, nb, ks)
    got = mem.to_host(ks)
    for b in range(nb):
        assert (got[b] == orc.pasta_keystream(t, key, 3 + b)).all()
    for e in golden["encrypt"]:
        if e["t"] != t:
            continue
        n = e["n"]
        S = 3  # every record restarts at counter 0: identical ciphertext rows for identical plaintext rows
        pt = np.tile(np.array([(7 * i + 3) % 256 for i in range(n)], dtype=np.uint64), (S, 1))
        d_in, d_out = mem.to_dev(pt), mem.empty((S, n))
        X.plain_crypt(key, d_in, S, n, d_out)
        ct = mem.to_host(d_out)
        for s in range(S):
            assert [int(v) for v in ct[s]] == e["ct"]
        X.plain_crypt(key, d_out, S, n, d_out, decrypt=True)  # in place
        assert (mem.to_host(d_out) == pt).all()


def check_decrypt(X, S, mem, B=3, seed=0):
    """Batched Decryptor::decrypt + BatchEncoder::decode vs the oracle, on fresh and on evaluated ciphertexts."""
    O = S.O
    rng = np.random.default_rng(seed)
    vals = rng.integers(0, O.t, (B, O.n), dtype=np.uint64)
    cts = np.stack([O.encrypt(S.pk, O.encode(vals[b]), 30 + b) for b in range(B)])
    # one evaluated ciphertext (rotation: key-switch noise) so that the rounding path is not trivial
    cts[B - 1] = O.rotate_rows(cts[B - 1], -1, S.gk)[0]
    out = mem.empty((B, O.n))
    X.decrypt(S.sk, mem.to_dev(cts), B, out)
    got = mem.to_host(out)
    for b in range(B):
        assert (got[b] == O.decode(O.decrypt(S.sk, cts[b]))).all()
    assert (got[0] == vals[0]).all()


def check_fc_variants(make_ctx, S, orc, mem, monkeypatch, n_in=37):
    """hhe_fc_row execution variants (per-child digit transforms, shared digits, forced exact fallback, with and without
    leaf sums) all return the oracle's words."""
    O = S.O
    rng = np.random.default_rng(8)
    v, w = rng.integers(0, 4, n_in), rng.integers(-8, 9, n_in)
    wc = O.encrypt(S.pk, O.encode(w), 43)
    B = 3
    vi = np.stack([O.encrypt(S.pk, O.encode(v), 41 + b) for b in range(B)])
    refs = [O.fc_row(vi[b], wc, S.rk, S.gk, n_in)[0] for b in range(B)]
    # (shared digits, leaf sums, items per chunk): chunk 1 -> three chunks round-robin over the internal streams
    for shared, leafsum, chunk in (("1", "1", "40"), ("0", "1", "40"), ("2", "1", "40"), ("1", "0", "40"), ("0", "0", "40"),
                                   ("1", "1", "1"), ("2", "1", "1"), ("0", "1", "2")):
        monkeypatch.setenv("HHE_FC_SHARED", shared)
        monkeypatch.setenv("HHE_FC_LEAFSUM", leafsum)
        monkeypatch.setenv("HHE_FC_CHUNK", chunk)
        X = make_ctx()
        S.load_keys(X)
        out = mem.empty((B,) + O.ct_shape)
        X.fc_row(mem.to_dev(vi), mem.to_dev(wc[None]), 1, n_in, out, B, relin_slot=0, default_galois_only=False)
        got = mem.to_host(out)
        for b in range(B):
            assert (got[b] == refs[b]).all(), (shared, leafsum, b)
        assert X.query("fc_fallbacks") == ((B + int(chunk) - 1) // int(chunk) if shared == "2" else 0)
        X.close()


def check_two_layer_chain(X, S, mem, n_in=24, seed=3, w1_vals=None, x_vals=None):
    """BASELINE config 4 shape (FC -> packed_square -> FC, SEAL_Cipher.cpp:547-552 between two sealhelper FC rows):
    ciphertext parity of the whole chain against the oracle's op sequence.  w1_vals / x_vals: first-layer weights (signed
    integers, encoded mod t as the analyst does) and inputs; defaults are small seeded values."""
    O = S.O
    rng = np.random.default_rng(seed)
    x = np.asarray(x_vals if x_vals is not None else rng.integers(0, 4, n_in), dtype=np.int64)
    w1i = np.asarray(w1_vals if w1_vals is not None else rng.integers(0, 4, n_in), dtype=np.int64)
    assert len(x) == n_in and len(w1i) == n_in
    vi = O.encrypt(S.pk, O.encode(x % S.t), 61)
    w1 = O.encrypt(S.pk, O.encode(w1i % S.t), 62)
    w2 = O.encrypt(S.pk, O.encode(rng.integers(0, 4, n_in)), 63)
    ref1, _ = O.fc_row(vi, w1, S.rk, S.gk, n_in)
    ref_sq = O.relinearize(O.multiply(ref1, ref1), S.rk)
    ref2, _ = O.fc_row(ref_sq, w2, S.rk, S.gk, n_in)
    d1, d3, dsq, d2 = mem.empty((1,) + O.ct_shape), mem.empty((1, 3, O.L, O.n)), mem.empty((1,) + O.ct_shape), mem.empty((1,) + O.ct_shape)
    X.fc_row(mem.to_dev(vi[None]), mem.to_dev(w1[None]), 1, n_in, d1, 1, relin_slot=0, default_galois_only=False)
    X.multiply(d1, d1, d3, 1)
    X.relinearize(d3, dsq, 1)
    X.fc_row(dsq, mem.to_dev(w2[None]), 1, n_in, d2, 1, relin_slot=0, default_galois_only=False)
    assert (mem.to_host(d1)[0] == ref1).all()
    assert (mem.to_host(dsq)[0] == ref_sq).all()
    assert (mem.to_host(d2)[0] == ref2).all()
    if O.noise_budget(S.sk, ref1, 8) > 0:  # the first layer decrypts to the plain integer dot product (FC == matMul, hhe_pktnn_examples.cpp:692-699)
        got = int(O.decode(O.decrypt(S.sk, mem.to_host(d1)[0]))[n_in - 1])
        assert got == int(np.dot(x, w1i)) % S.t
