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
