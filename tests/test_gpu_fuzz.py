"""Seeded differential sweep on the GPU: random parameter sets (degree, number and size of the primes) through the op
checks, a ragged transciphering and an FC row, all bit for bit against the oracle."""
import numpy as np
import pytest

from conftest import Setup
import parity_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mem():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return pc.TorchMem("cuda:0")


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_random_parameter_sets(orc, api, mem, seed):
    lib = api.load_library()
    rng = np.random.default_rng(seed)
    logn = int(rng.integers(10, 13))
    K = int(rng.integers(3, 7))
    bits = [int(b) for b in rng.integers(42, 61, K)]
    bits[-1] = max(bits)  # SEAL: the special prime is at least as large as the data primes
    S = Setup(orc, logn, bits, all_galois=True)
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    pc.check_ntt(X, S.O, mem, seed=seed)
    pc.check_ops(X, S, mem, B=2, seed=seed)
    nwords = int(rng.integers(1, 300))
    pt = [int(v) for v in rng.integers(0, 256, nwords)]
    cw, ncw = S.sym_blocks(orc, pt)
    nb = len(ncw)
    out = mem.empty((nb,) + S.O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, np.arange(nb), out)
    res = mem.to_host(out)
    for b in range(nb):
        assert (res[b] == S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b)).all(), (seed, b)
    n_in = int(rng.integers(2, (1 << logn) // 2))
    vi = S.O.encrypt(S.pk, S.O.encode(rng.integers(0, 4, n_in)), 5)
    wc = S.O.encrypt(S.pk, S.O.encode(rng.integers(0, 8, n_in)), 6)
    o = mem.empty((1,) + S.O.ct_shape)
    X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, o, 1, relin_slot=0, default_galois_only=False)
    assert (mem.to_host(o)[0] == S.O.fc_row(vi, wc, S.rk, S.gk, n_in)[0]).all(), (seed, n_in)


@pytest.mark.parametrize("seed,logn", [(505, 12), (606, 13)])
def test_random_parameter_sets_full_tiles(orc, api, mem, seed, logn):
    """the same sweep at N >= 4096, where the matmul loop and every generic key switch run through ks_row_kernel
    (L up to 6: the lazy sums are folded after the fourth digit)"""
    lib = api.load_library()
    rng = np.random.default_rng(seed)
    K = int(rng.integers(5, 8))
    bits = [int(b) for b in rng.integers(44, 61, K)]
    bits[-1] = max(bits)
    S = Setup(orc, logn, bits, all_galois=True)
    X = api.Context(S.logn, S.q, S.t, lib=lib)
    S.load_keys(X)
    pc.check_ops(X, S, mem, B=3, seed=seed)
    nwords = int(rng.integers(129, 257))
    pt = [int(v) for v in rng.integers(0, 256, nwords)]
    cw, ncw = S.sym_blocks(orc, pt)
    out = mem.empty((2,) + S.O.ct_shape)
    X.transcipher(mem.to_dev(S.enc_key), cw, ncw, np.arange(2), out)
    res = mem.to_host(out)
    for b in range(2):
        assert (res[b] == S.O.transcipher_block(S.enc_key, S.rk, S.gk, cw[b, :ncw[b]], b)).all(), (seed, b)
    n_in = int(rng.integers(2, 48))
    vi = S.O.encrypt(S.pk, S.O.encode(rng.integers(0, 4, n_in)), 5)
    wc = S.O.encrypt(S.pk, S.O.encode(rng.integers(0, 8, n_in)), 6)
    o = mem.empty((1,) + S.O.ct_shape)
    X.fc_row(mem.to_dev(vi[None]), mem.to_dev(wc[None]), 1, n_in, o, 1, relin_slot=0, default_galois_only=False)
    assert (mem.to_host(o)[0] == S.O.fc_row(vi, wc, S.rk, S.gk, n_in)[0]).all(), (seed, n_in)
