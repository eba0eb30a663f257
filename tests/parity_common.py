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
    # fourth column: leaf key switches per launch (leaves of any nodes whose digit transforms are resident; 1 = one leaf at a time)
    for shared, leafsum, chunk, group in (("1", "1", "40", "4"), ("0", "1", "40", "4"), ("2", "1", "40", "4"), ("1", "0", "40", "4"),
                                          ("0", "0", "40", "4"), ("1", "1", "1", "4"), ("2", "1", "1", "4"), ("0", "1", "2", "4"),
                                          ("1", "1", "40", "1"), ("1", "1", "2", "2"), ("1", "1", "40", "3")):
        monkeypatch.setenv("HHE_FC_LEAFGROUP", group)
        monkeypatch.setenv("HHE_FC_SHARED", shared)
        monkeypatch.setenv("HHE_FC_LEAFSUM", leafsum)
        monkeypatch.setenv("HHE_FC_CHUNK", chunk)
        X = make_ctx()
        S.load_keys(X)
        out = mem.empty((B,) + O.ct_shape)
        X.fc_row(mem.to_dev(vi), mem.to_dev(wc[None]), 1, n_in, out, B, relin_slot=0, default_galois_only=False)
        got = mem.to_host(out)
        for b in range(B):
            assert (got[b] == refs[b]).all(), (shared, leafsum, chunk, group, b)
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


def check_key_sets(X, S, orc, mem, threads=True):
    """Key objects with identity (CSP.cpp:238-242, 271-278, 306, 312-316; Analyst.cpp:62-94): two GaloisKeys objects and two
    RelinKeys objects under ONE secret key, made with different randomness.  A = every default element (what
    create_galois_keys() without arguments makes), B = {0, -1, 128, -128, -256, -384}.  Every call must produce the words the
    oracle produces WITH THE SET THE CALL NAMES: rotate_rows(-384) NAF-decomposes over A and is one key switch with B; flatten
    differs between A and B; the FC uses (rk2, A); nothing uploaded into one set is seen through another."""
    O = S.O
    n = O.n
    eltsA = list(dict.fromkeys(int(e) for e in O.galois_elts_all()))
    stepsB = [0, -1, 128, -128, -256, -384]
    eltsB = list(dict.fromkeys(int(O.galois_elt(s)) for s in stepsB))
    gkA = O.keygen_galois(S.sk, eltsA, 101)
    gkB = O.keygen_galois(S.sk, eltsB, 202)   # same elements where they overlap, different randomness: different words
    rk2 = O.keygen_relin(S.sk, 303)
    A, Bs, R2 = X.keyset(), X.keyset(), X.keyset()
    for e, k in zip(gkA.elts, gkA.keys):
        A.set_galois(int(e), k)
    for e, k in zip(gkB.elts, gkB.keys):
        Bs.set_galois(int(e), k)
    R2.set_relin(rk2)
    A.set_relin(S.rk)  # a set may hold both kinds
    assert A.has_galois(O.galois_elt(-128)) and Bs.has_galois(O.galois_elt(-384)) and not A.has_galois(O.galois_elt(-384))
    assert R2.has_relin() and not Bs.has_relin()
    rng = np.random.default_rng(77)
    cts = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, O.t, n)), 70 + b) for b in range(2)])
    d, out = mem.to_dev(cts), mem.empty(cts.shape)
    # rotate_rows(-384): NAF over A ({128, -512} at N = 2^15; the +-N/2 term is skipped when it is a whole row), direct with B
    refA = [O.rotate_rows(cts[b], -384, gkA) for b in range(2)]
    refB = [O.rotate_rows(cts[b], -384, gkB) for b in range(2)]
    assert n >= 2048 and refB[0][1] == 1 and refA[0][1] == 2   # (at N = 1024 a row has 512 slots and -384 is the element of +128)
    X.rotate_rows(d, -384, out, 2, gk=A)
    hA = mem.to_host(out)
    X.rotate_rows(d, -384, out, 2, gk=Bs)
    hB = mem.to_host(out)
    for b in range(2):
        assert (hA[b] == refA[b][0]).all() and (hB[b] == refB[b][0]).all()
    assert not (hA[0] == hB[0]).all()
    # an element both sets hold: same rotation, different key words -> different ciphertext words, each equal to its oracle
    for ks, gk in ((A, gkA), (Bs, gkB)):
        X.rotate_rows(d, -128, out, 2, gk=ks)
        assert (mem.to_host(out)[1] == O.rotate_rows(cts[1], -128, gk)[0]).all()
        X.rotate_columns(d, out, 2, gk=ks)
        assert (mem.to_host(out)[0] == O.rotate_columns(cts[0], gk)).all()
    # the default set is empty: the un-named call fails, and a step neither direct nor NAF-reachable in B fails with B
    for kw in ({}, {"gk": Bs}):
        try:
            X.rotate_rows(d, -5, out, 2, **kw)
            raise AssertionError("rotation without its key must fail")
        except RuntimeError as e:
            assert "Galois key not present" in str(e)
    # flatten(in, out, galois_keys): 3 blocks, steps -128, -256 -- direct in B, direct in A as well but with A's words
    blocks = np.stack([O.encrypt(S.pk, O.encode(rng.integers(0, O.t, 128)), 80 + i) for i in range(3)])
    fo = mem.empty((1,) + O.ct_shape)
    X.flatten(mem.to_dev(blocks[None]), 3, fo, 1, gk=A)
    fA = mem.to_host(fo)[0]
    X.flatten(mem.to_dev(blocks[None]), 3, fo, 1, gk=Bs)
    fB = mem.to_host(fo)[0]
    assert (fA == O.flatten(blocks, gkA)).all() and (fB == O.flatten(blocks, gkB)).all() and not (fA == fB).all()
    # relinearize / FC with the objects the CSP names: csp rk (R2) and the analyst's default Galois keys (A)
    o3 = mem.empty((2, 3) + O.ct_shape[1:])
    X.multiply(d, d, o3, 2)
    X.relinearize(o3, out, 2, rk=R2)
    assert (mem.to_host(out)[0] == O.relinearize(O.multiply(cts[0], cts[0]), rk2)).all()
    X.relinearize(o3, out, 2, rk=A)
    assert (mem.to_host(out)[0] == O.relinearize(O.multiply(cts[0], cts[0]), S.rk)).all()
    n_in = 45
    v, w = rng.integers(0, 4, n_in), rng.integers(-8, 9, n_in)
    vi, wc = O.encrypt(S.pk, O.encode(v), 91), O.encrypt(S.pk, O.encode(w % O.t), 92)
    X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, fo, 1, rk=R2, gk=A)
    ref_fc, _ = O.fc_row(vi, wc, rk2, gkA, n_in)
    assert (mem.to_host(fo)[0] == ref_fc).all()
    if not threads:
        return
    # two request handlers at once (CSPRPC.cpp:201-203), each naming its own set on the shared context
    import threading
    res, errs = {}, []

    def work(name, ks, step):
        try:
            o = mem.empty(cts.shape)
            for _ in range(3):
                X.rotate_rows(mem.to_dev(cts), step, o, 2, gk=ks)
            res[name] = mem.to_host(o)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=("A", A, -128)), threading.Thread(target=work, args=("B", Bs, -128))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert (res["A"][0] == O.rotate_rows(cts[0], -128, gkA)[0]).all() and (res["B"][0] == O.rotate_rows(cts[0], -128, gkB)[0]).all()
    for ks in (A, Bs, R2):
        ks.close()


def check_matmul_adversarial(X, S, orc, mem, pattern):
    """The matmul loop on worst-case residues (the lazy ranges are tightest at 60-bit primes, 16q ~ 2^64): the 'ciphertext' the
    loop starts from, the symmetric words and -- for 'max_keys' -- every key-switch key word sit at q_j - 1 (or alternate with
    0).  Nothing here is a valid encryption; the oracle's exact arithmetic defines the expected words all the same."""
    O = S.O
    enc = np.zeros(O.ct_shape, np.uint64)
    for j in range(O.L):
        if pattern == "alt":
            enc[:, j, ::2] = S.q[j] - 1
        else:
            enc[:, j, :] = S.q[j] - 1
    rk, gk = S.rk, S.gk
    if pattern == "max_keys":
        rk = np.zeros_like(S.rk)
        for j in range(O.K):
            rk[:, :, j, :] = S.q[j] - 1
        gk = orc.GaloisKeys(S.gk.elts, np.stack([rk] * len(S.gk.elts)))
    ks = X.keyset()
    ks.set_relin(rk)
    for e, k in zip(gk.elts, gk.keys):
        ks.set_galois(int(e), k)
    cw = np.full((1, 128), S.t - 1, np.uint64)
    out = mem.empty((1,) + O.ct_shape)
    X.transcipher(mem.to_dev(enc), cw, [128], [0], out, rk=ks, gk=ks)
    assert (mem.to_host(out)[0] == O.transcipher_block(enc, rk, gk, cw[0], 0)).all(), pattern
    ks.close()
