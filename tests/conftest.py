import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "privacy-preserving-ml-through-hhe_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pkg_api():
    return importlib.import_module(PKG + ".api")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def api():
    return pkg_api()


@pytest.fixture(scope="session")
def emu_lib(api):
    """tests-only CPU emulator of the kernel bodies + the real host driver (tests/emu)."""
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    return api.load_library(os.path.join(d, "libhhe_emu.so"))


T = 65537
DEMO_KEY = np.array([(i * 2654435761 + 12345) % T for i in range(256)], dtype=np.uint64)


class Setup:
    """A BFV context with keys, a PASTA key and its BFV encryption, all from the oracle."""

    def __init__(self, orc, logn, bits, t=T, all_galois=False, extra_steps=()):
        self.t, self.logn, self.n = t, logn, 1 << logn
        self.q = orc.coeff_modulus_create(self.n, bits)
        self.O = O = orc.Oracle(logn, self.q, t)
        self.sk = O.keygen_secret(1)
        self.pk = O.keygen_public(self.sk, 2)
        self.rk = O.keygen_relin(self.sk, 3)
        steps = [-1, 0] + ([128] if self.n // 2 != 128 else []) + list(extra_steps)
        elts = O.galois_elts_all() if all_galois else [O.galois_elt(s) for s in steps]
        if all_galois:
            for s in steps:
                if O.galois_elt(s) not in elts:
                    elts.append(O.galois_elt(s))
        elts = list(dict.fromkeys(int(e) for e in elts))  # SEAL keeps one key per element (get_elts_all repeats 3^(N/4))
        self.gk = O.keygen_galois(self.sk, elts, 7)
        self.key = np.array([(i * 2654435761 + 12345) % t for i in range(256)], dtype=np.uint64)
        self.enc_key = O.encrypt(self.pk, O.pasta_pack_key(self.key), 11)

    def load_keys(self, X):
        X.set_relin_key(self.rk)
        for i, e in enumerate(self.gk.elts):
            X.set_galois_key(int(e), self.gk.keys[i])

    def sym_blocks(self, orc, pt):
        """PASTA-encrypt pt and split into [nb][128] words + counts."""
        pt = np.asarray(pt, dtype=np.uint64)
        sym = orc.pasta_encrypt(self.t, self.key, pt)
        nb = (len(pt) + 127) // 128
        cw = np.zeros((nb, 128), np.uint64)
        ncw = np.zeros(nb, np.uint32)
        for b in range(nb):
            seg = sym[b * 128:(b + 1) * 128]
            cw[b, :len(seg)] = seg
            ncw[b] = len(seg)
        return cw, ncw


@pytest.fixture(scope="session")
def small(orc):
    """N=1024, 9 x 50-bit primes: PASTA-3 transciphering decrypts correctly (budget ~135 bits)."""
    return Setup(orc, 10, [50] * 9)
