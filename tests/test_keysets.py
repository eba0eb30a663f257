"""Key objects with identity at the C ABI (hhe_keyset): the reference's CSP holds several RelinKeys / GaloisKeys objects and
names one per call (src/examples/CSP/CSP.cpp:238-242, 271-278, 306, 312-316; created at src/examples/Analyst/Analyst.cpp:62-94).
CPU part: the product's host driver on the tests-only emulator, against the oracle called with the same set."""
import numpy as np
import pytest

from conftest import Setup
import parity_common as pc


@pytest.fixture(scope="module")
def mem():
    return pc.HostMem()


def test_two_galois_sets_and_two_relin_sets_under_one_secret_key(orc, api, emu_lib, mem):
    S = Setup(orc, 11, [50] * 4)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)   # default set left empty on purpose
    pc.check_key_sets(X, S, orc, mem)


def test_key_sets_full_tiles_row_kernel(orc, api, emu_lib, mem):
    # N = 4096: rotations / relinearize run through the fused row kernel and its per-key Shoup tables (keyed by key, not by element)
    S = Setup(orc, 12, [50] * 3)
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    pc.check_key_sets(X, S, orc, mem, threads=False)


def test_decompose_names_three_key_objects(orc, api, emu_lib, mem):
    """BaseCSP::decompose: PASTA_SEAL(.., analyst rk, analyst gk).decomposition + flatten(.., csp gk): transciphering keys from
    one object, flatten keys from another with different words for the shared element -128."""
    S = Setup(orc, 10, [50] * 9)
    O = S.O
    X = api.Context(S.logn, S.q, S.t, lib=emu_lib)
    pasta, flat = X.keyset(), X.keyset()
    pasta.set_relin(S.rk)
    for e, k in zip(S.gk.elts, S.gk.keys):
        pasta.set_galois(int(e), k)
    gk_flat = O.keygen_galois(S.sk, [int(O.galois_elt(s)) for s in (0, -1, 128, -128, -256)], 555)
    for e, k in zip(gk_flat.elts, gk_flat.keys):
        flat.set_galois(int(e), k)
    nwords = 300
    pt = np.array([(7 * i + 3) % 256 for i in range(nwords)], dtype=np.uint64)
    rec = orc.pasta_encrypt(S.t, S.key, pt)
    out = mem.empty((1,) + O.ct_shape)
    X.decompose(mem.to_dev(S.enc_key), rec[None], out, mask_last=True, rk=pasta, gk=pasta, flatten_gk=flat)
    cw, ncw = S.sym_blocks(orc, pt)
    blocks = [O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b) for b in range(3)]
    blocks[2] = O.mask(blocks[2], np.ones(44, np.uint64))
    ref = O.flatten(np.stack(blocks), gk_flat)
    got = mem.to_host(out)[0]
    assert (got == ref).all()
    assert (O.decode(O.decrypt(S.sk, got))[:nwords] == pt).all()
    # the PASTA set has no key for -128 / -256 and they are not NAF-reachable through {-1, 128, columns}: naming it for flatten fails
    with pytest.raises(api.HheError) as e:
        X.decompose(mem.to_dev(S.enc_key), rec[None], out, mask_last=True, rk=pasta, gk=pasta, flatten_gk=pasta)
    assert e.value.code == api.ERR_NO_GALOIS_KEY


def test_key_set_errors(orc, api, emu_lib, mem, small):
    X = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    Y = api.Context(small.logn, small.q, small.t, lib=emu_lib)
    ks = X.keyset()
    bad = small.rk.copy()
    bad[0, 0, 0, 0] = small.q[0]   # not reduced modulo its prime: the Shoup-quotient products would be wrong
    with pytest.raises(api.HheError) as e:
        ks.set_relin(bad)
    assert e.value.code == api.ERR_INVALID and not ks.has_relin()
    with pytest.raises(api.HheError):
        ks.set_galois(4, small.rk)  # even: not a Galois element
    with pytest.raises(api.HheError) as e:
        X.set_galois_key(3, bad)
    assert e.value.code == api.ERR_INVALID and not X.has_galois_key(3)
    ct = mem.to_dev(small.enc_key[None])
    out = mem.empty((1,) + small.O.ct_shape)
    ks.set_galois(int(small.O.galois_elt(-1)), small.gk.keys[list(small.gk.elts).index(small.O.galois_elt(-1))])
    with pytest.raises(api.HheError) as e:
        Y.rotate_rows(ct, -1, out, 1, gk=ks)   # a set of another context
    assert e.value.code == api.ERR_INVALID
    X.rotate_rows(ct, -1, out, 1, gk=ks)
    assert (mem.to_host(out)[0] == small.O.rotate_rows(small.enc_key, -1, small.gk)[0]).all()
    # replacing a key inside a set replaces what was derived from it
    other = small.O.keygen_galois(small.sk, [int(small.O.galois_elt(-1))], 999)
    ks.set_galois(int(small.O.galois_elt(-1)), other.keys[0])
    X.rotate_rows(ct, -1, out, 1, gk=ks)
    assert (mem.to_host(out)[0] == small.O.rotate_rows(small.enc_key, -1, other)[0]).all()
    ks.close()
    X.close()   # destroying the context after (and, for forgotten sets, instead of) its sets
    ks2 = Y.keyset()
    Y.close()
    ks2.h = None
