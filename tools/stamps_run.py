#!/usr/bin/env python3
"""Diagnostic: one config-2 transciphering batch through tools/libhhe_stamps.so, then the per-phase timeline (shader cycles) of the
sampled ks_row_kernel workgroups of the LAST launch.  Never part of the product or of bench.py's numbers."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HHE_LIB"] = os.path.join(ROOT, "tools", "libhhe_stamps.so")
import torch
import bench
api = importlib.import_module(bench.PKG + ".api")
lib = api.load_library()
q, t, logn = bench.Q_CONFIG2, bench.T_PLAIN, 15
n, K, L = 1 << logn, len(q), len(q) - 1
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
X = api.Context(logn, q, t, device=0, lib=lib)
rng = np.random.default_rng(1)
for step in (-1, 128, 0):
    X.set_galois_key(X.query("galois_elt", step), bench.synthetic_keys(rng, q, n))
X.set_relin_key(bench.synthetic_keys(rng, q, n))
d_key = torch.from_numpy(bench.synthetic_ct(rng, q, n).view(np.int64)).cuda()
cw = rng.integers(0, t, size=(B, 128), dtype=np.uint64)
out = torch.zeros((B, 2, L, n), dtype=torch.int64, device="cuda")
for _ in range(2):
    X.transcipher(d_key, cw, np.full(B, 128, np.uint32), np.zeros(B, np.uint64), out)
torch.cuda.synchronize()
SL, WG = 24, 4096
buf = np.zeros(SL * WG, np.uint64)
lib.hhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.hhe_debug_read_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(WG, SL).astype(np.int64)
if len(sys.argv) > 2:
    np.save(sys.argv[2], s)
reps = int(os.environ.get('HHE_KS_REPS', '2'))
nmain = B * K   # sampled key-switch workgroups: every 64th of B*K*64 tiles
main = s[:min(nmain, WG)]
main = main[(main[:, 0] > 0) & (main[:, 16] > main[:, 0])]
names = {1: "twiddle fill"}
for I in range(3):
    names[2 + 3 * I] = f"digit {I}: tile load + LDS write"
    names[3 + 3 * I] = f"digit {I}: forward rounds"
    names[4 + 3 * I] = f"digit {I}: key products"
names.update({11: "flush 1", 12: "inverse rounds 1", 13: "store 1 / S_0 store", 14: "flush 2", 15: "inverse rounds 2", 16: "store 2"})
for kind, sel in (("data limbs (J < L)", main[:, 11] == 0), ("special limb", main[:, 11] != 0)):
    m = main[sel]
    if not len(m):
        continue
    print(f"--- {kind}: {len(m)} workgroups, total {np.median(m[:, 16] - m[:, 0]):.0f} cycles (median), p10 {np.percentile(m[:, 16] - m[:, 0], 10):.0f}, p90 {np.percentile(m[:, 16] - m[:, 0], 90):.0f}")
    prev = 0
    for i in sorted(names):
        if (m[:, i] == 0).all():
            continue
        d = m[:, i] - m[:, prev]
        print(f"  {names[i]:34s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
        prev = i
span = main[:, 16].max() - main[:, 0].min()
print(f"launch span (first start to last end among sampled) {span} cycles; sum of per-wg totals / span = {(main[:, 16] - main[:, 0]).sum() / span:.1f} sampled wgs in flight on average (x64 for all)")
# where the short c0 tiles (last in the grid) sit in the launch: their share of the kernel's duration (the library records ONE
# launch, HHE_STAMP_LAUNCH, default the 300th ks_row launch of the process: a middle step of the second call)
c0 = s[min(nmain, WG):]
c0 = c0[(c0[:, 0] > 0) & (c0[:, 16] > c0[:, 0])]
if len(c0):
    t0 = min(main[:, 0].min(), c0[:, 0].min())
    print(f"{len(main)} key-switch workgroups sampled: start {main[:, 0].min() - t0}..{main[:, 0].max() - t0}, last end {main[:, 16].max() - t0}")
    print(f"{len(c0)} c0 tiles sampled: life median {np.median(c0[:, 16] - c0[:, 0]):.0f} cycles (p10 {np.percentile(c0[:, 16] - c0[:, 0], 10):.0f}, p90 {np.percentile(c0[:, 16] - c0[:, 0], 90):.0f}), "
          f"start {c0[:, 0].min() - t0}..{c0[:, 0].max() - t0} (median {np.median(c0[:, 0]) - t0:.0f}), last end {c0[:, 16].max() - t0}")
    e = np.sort(np.concatenate([main[:, 16], c0[:, 16]])) - t0
    print("ends of all sampled workgroups, deciles:", [int(v) for v in np.percentile(e, range(0, 101, 10))])
