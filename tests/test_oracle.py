"""Pin the CPU oracle: SURVEY Appendix A known answers, the reference-built plain PASTA-3
golden vectors (tests/golden/pasta_plain.json), and the reference's own end-to-end properties
(hhe_pktnn_examples.cpp:639-648 decrypt(transcipher(c)) == plaintext; :692-699 FC == plain matmul)."""
import hashlib
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T = 65537


def test_survey_a10_a1_a7_parameter_derivation(orc):
    q = orc.coeff_modulus_create(32768, [60, 60, 60, 60])
    assert q == [1152921504595968001, 1152921504597016577, 1152921504598720513, 1152921504606584833]
    assert orc.minimal_primitive_root(65536, q[0]) == 88651361085495
    O = orc.Oracle(15, q, T)
    assert [O.query("bsk", i) for i in range(3)] == [2305843009211400193, 2305843009210023937, 2305843009208713217]
    assert O.query("bsk", 3) == 2305843009211662337
    assert O.query("gamma") == 2305843009211596801
    # plain-modulus NTT root for t=65537, N=2^15 is 3: psi^bitrev(1)... table[1] = psi^(N/2)
    assert orc.minimal_primitive_root(65536, T) == 3


def test_survey_a3_galois_and_naf(orc):
    O = orc.Oracle(15, orc.coeff_modulus_create(32768, [60, 60]), T)
    assert O.galois_elt(-1) == 43691 and O.galois_elt(128) == 31233
    assert O.galois_elt(-128) == 34305 and O.galois_elt(0) == 65535
    assert len(O.galois_elts_all()) == 29
    assert orc.naf(-783) == [1, -16, 256, -1024]
    assert orc.naf(-299) == [1, 4, 16, -64, -256]
    assert orc.naf(128) == [128] and orc.naf(3) == [-1, 4]


def test_bfv_default_16384_are_valid_ntt_primes(orc):
    # SURVEY A.10 (reference default N=16384 / BFVDefault)
    dflt = [281474976546817, 281474976317441, 281474975662081, 562949952798721, 562949952700417,
            562949952274433, 562949951979521, 562949951881217, 562949951619073]
    for p in dflt:
        assert orc.lib().orc_is_prime(p) and (p - 1) % 32768 == 0


def test_shake128_matches_hashlib(orc):
    for n in (0, 1, 16, 167, 168, 169, 500):
        d = bytes((7 * i + 1) & 0xFF for i in range(n))
        assert orc.shake128(d, 777) == hashlib.shake_128(d).digest(777)


def test_survey_a8_pasta_known_answers(orc):
    mats, rcs = orc.pasta_block_randomness(T, 0)
    assert list(mats[0, 0, 0, :3]) == [34686, 37780, 45807]
    assert list(mats[0, 0, 1, :2]) == [8576, 58655]
    assert mats[0, 0, 127, 127] == 55028 and mats[0, 1, 0, 0] == 14178
    assert rcs[0, 0, 0] == 30715 and rcs[0, 0, 127] == 62165 and rcs[0, 1, 0] == 1318 and rcs[0, 1, 127] == 38851
    key = [(i * 2654435761 + 12345) % T for i in range(256)]
    pt = [(7 * i + 3) % 256 for i in range(300)]
    ct = orc.pasta_encrypt(T, key, pt)
    assert list(ct[:4]) == [38641, 24494, 19771, 27070] and ct[128] == 46700 and ct[299] == 51147
    assert list(orc.pasta_decrypt(T, key, ct)) == pt


def test_reference_built_pasta_golden(orc):
    g = json.load(open(os.path.join(HERE, "golden", "pasta_plain.json")))
    for c in g["randomness"]:
        t = c["t"]
        mats, rcs = orc.pasta_block_randomness(t, c["block"], c["nonce"])
        assert hashlib.sha256(mats.tobytes()).hexdigest() == c["mats_sha256"]
        assert hashlib.sha256(rcs.tobytes()).hexdigest() == c["rcs_sha256"]
        key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
        assert [int(v) for v in orc.pasta_keystream(t, key, c["block"], c["nonce"])] == c["keystream"]
    for e in g["encrypt"]:
        t = e["t"]
        key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
        pt = [(7 * i + 3) % 256 for i in range(e["n"])]
        assert [int(v) for v in orc.pasta_encrypt(t, key, pt)] == e["ct"]


def test_ntt_roundtrip_and_negacyclic_convolution(orc):
    O = orc.Oracle(10, orc.coeff_modulus_create(1024, [50, 50]), T)
    rng = np.random.default_rng(3)
    q = O.q[0]
    a = rng.integers(0, q, O.n, dtype=np.uint64)
    b = np.zeros(O.n, np.uint64)
    b[1] = 1  # multiply by x: negacyclic shift
    assert (O.ntt_inv(0, O.ntt_fwd(0, a)) == a).all()
    prod = [(int(x) * int(y)) % q for x, y in zip(O.ntt_fwd(0, a), O.ntt_fwd(0, b))]
    c = O.ntt_inv(0, np.array(prod, dtype=np.uint64))
    exp = np.roll(a, 1)
    exp[0] = (q - int(a[-1])) % q
    assert (c == exp).all()


def test_bfv_ops_decrypt_correctly(orc, small):
    O, sk, pk, gk = small.O, small.sk, small.pk, small.gk
    n, t, h = O.n, T, O.n // 2
    rng = np.random.default_rng(0)
    v, w = rng.integers(0, t, n), rng.integers(0, t, n)
    pv, pw = O.encode(v), O.encode(w)
    assert (O.decode(pv) == v).all()
    ct = O.encrypt(pk, pv, 5)
    dec = lambda c: O.decode(O.decrypt(sk, c))
    assert (dec(ct) == v).all()
    assert (dec(O.encrypt_symmetric(sk, pv, 6)) == v).all()
    assert (dec(O.multiply_plain(ct, pw)) == (v * w) % t).all()
    assert (dec(O.add_plain(ct, pw)) == (v + w) % t).all()
    assert (dec(O.sub_plain(ct, pw)) == (v - w) % t).all()
    assert (dec(O.negate(ct)) == (-v) % t).all()
    r, ks = O.rotate_rows(ct, -1, gk)
    assert ks == 1 and (dec(r) == np.concatenate([np.roll(v[:h], 1), np.roll(v[h:], 1)])).all()
    r, _ = O.rotate_rows(ct, 128, gk)
    assert (dec(r) == np.concatenate([np.roll(v[:h], -128), np.roll(v[h:], -128)])).all()
    assert (dec(O.rotate_columns(ct, gk)) == np.concatenate([v[h:], v[:h]])).all()
    ct2 = O.encrypt(pk, pw, 8)
    m3 = O.multiply(ct, ct2)
    assert (dec(m3) == (v * w) % t).all()
    assert (O.multiply(ct2, ct) == m3).all()  # operand-order independent
    assert (dec(O.relinearize(m3, small.rk)) == (v * w) % t).all()
    assert O.noise_budget(sk, ct) > 300


def test_transcipher_decrypts_to_plaintext(orc, small):
    """the reference's own oracle for the path: hhe_pktnn_examples.cpp:639-648"""
    O = small.O
    pt = np.array([(7 * i + 3) % 256 for i in range(300)], dtype=np.uint64)
    cw, ncw = small.sym_blocks(orc, pt)
    for b in (0, 2):
        out = O.transcipher_block(small.enc_key, small.rk, small.gk, cw[b, :ncw[b]], b)
        dec = O.decode(O.decrypt(small.sk, out))[:ncw[b]]
        assert (dec == pt[b * 128:b * 128 + ncw[b]]).all()
        assert O.noise_budget(small.sk, out) > 60
