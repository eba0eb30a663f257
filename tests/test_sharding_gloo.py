"""N>1 path on CPU: two gloo ranks shard independent samples, transcipher their share through the C ABI
(tests-only emulator backend here; RCCL ranks do the same on GPUs with no data-path collective) and the
union of the shards equals the single-process result."""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "privacy-preserving-ml-through-hhe_amd"

WORKER = r'''
import importlib, os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as orc
from conftest import Setup
sh = importlib.import_module(PKG + ".sharding"); api = importlib.import_module(PKG + ".api")
rank, world = sh.init_process_group("gloo")
lib = api.load_library(os.path.join(ROOT, "tests", "emu", "libhhe_emu.so"))
S = Setup(orc, 10, [50] * 9)
X = api.Context(S.logn, S.q, S.t, lib=lib); S.load_keys(X)
n_samples, words = 3, 200   # 2 blocks per sample (128 + 72)
lo, hi = sh.shard_samples(n_samples, rank, world)
items = sh.work_items(lo, hi, words)
cw = np.zeros((len(items), 128), np.uint64); ncw = []; bi = []
for i, (s, b, w) in enumerate(items):
    pt = np.array([(7 * j + 3 + s) % 256 for j in range(words)], dtype=np.uint64)
    sym = orc.pasta_encrypt(S.t, S.key, pt)
    cw[i, :w] = sym[b * 128:b * 128 + w]; ncw.append(w); bi.append(b)
out = np.zeros((len(items),) + S.O.ct_shape, np.uint64)
if len(items): X.transcipher(S.enc_key.copy(), cw, ncw, bi, out)
sh.barrier()
el, units = sh.reduce_max_sum(1.0 + rank, len(items))
np.save(os.path.join(OUT, f"shard{rank}.npy"), out)
if rank == 0: open(os.path.join(OUT, "agg.txt"), "w").write(f"{el} {units}")
'''


def test_two_gloo_ranks_cover_all_items(tmp_path, emu_lib, orc):
    sh = importlib.import_module(PKG + ".sharding")
    assert sh.shard_samples(10, 0, 4) == (0, 3) and sh.shard_samples(10, 3, 4) == (8, 10)
    assert sh.sample_blocks(784) == [(b, 128) for b in range(6)] + [(6, 16)]
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT={ROOT!r}\nPKG={PKG!r}\nOUT={str(tmp_path)!r}\n" + WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29533", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")
    assert a.shape[0] + b.shape[0] == 6  # 3 samples x 2 blocks, no overlap, none missing
    el, units = open(tmp_path / "agg.txt").read().split()
    assert float(el) == 2.0 and int(units) == 6  # MAX over ranks, SUM of units
    # single-process result for the same items
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Setup
    S = Setup(orc, 10, [50] * 9)
    both = np.concatenate([a, b])
    i = 0
    for s in range(3):
        pt = np.array([(7 * j + 3 + s) % 256 for j in range(200)], dtype=np.uint64)
        cw, ncw = S.sym_blocks(orc, pt)
        for blk in range(2):
            dec = S.O.decode(S.O.decrypt(S.sk, both[i]))[:ncw[blk]]
            assert (dec == pt[blk * 128:blk * 128 + ncw[blk]]).all()
            i += 1
