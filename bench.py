#!/usr/bin/env python3
"""bench.py -- PASTA-3 transcipherings/sec on MI355X (BASELINE.json metric).

step      = one pass of the hot path (hhe_pasta3_transcipher) over one batch of B independent
            128-word blocks per rank (BASELINE config 2: N=2^15, 4x60-bit primes, t=65537, B=256).
value     = transcipherings all ranks completed / max-over-ranks time (inputs resident in HBM;
            only the 1 KiB/block of symmetric ciphertext words crosses PCIe inside the timed region).
--gpus N  = N > 1 without a launcher: this process starts N rank processes itself (launch_ranks); under
            torch.distributed.run the ranks come from the environment.  Ranks shard by independent items,
            no data-path collective; RCCL only carries the barrier and the MAX/SUM of (time, units).
roofline  = the dominant kernel (ks_row_kernel): its average launch duration is measured live with HIP events on the
            stream it is launched on (hhe_ctx_profile: one more step of the same workload with every launch bracketed),
            against the bytes it must move; `path` repeats the same for the whole transciphering
            with SURVEY 8(d)'s op-level model (A_block).
cpu_baseline = the CPU oracle (a C port of the reference schedule) on the host cores, rank 0, N=1 only.
extras    = (N=1 only) BASELINE config 5: a 512-step rotate_rows chain at N=2^16 / 6 primes; config 3 shape: MNIST-shaped samples
            through hhe_decompose + the 784x10 FC; the reference's default parameters (N=2^14, BFVDefault 9 primes); the drop-in's
            per-record call shape (one 784-word / one 300-word record per hhe_decompose call, ms per call).
--workload mnist-e2e = BASELINE config 3 as its own bench: every rank runs hhe_decompose + the 10 FC rows on ITS contiguous
            sample range (sharding.shard_samples), value = samples/s (SUM of samples / MAX of time), decompose and FC rooflines.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "privacy-preserving-ml-through-hhe_amd"

# BASELINE config 2: CoeffModulus::Create(32768, {60,60,60,60})
Q_CONFIG2 = [1152921504595968001, 1152921504597016577, 1152921504598720513, 1152921504606584833]
# BASELINE config 5: CoeffModulus::Create(65536, {60 x 6}) (tests/test_abi.py checks the list against the oracle's Create)
Q_CONFIG5 = [1152921504592429057, 1152921504592822273, 1152921504595968001, 1152921504597016577, 1152921504598720513,
             1152921504606584833]
T_CONFIG5 = 8088322049  # src/configs/config.cpp:22 (65537 cannot batch at N = 2^16)
T_PLAIN = 65537
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
PMC_PROFILE = os.path.join(ROOT, "profiles", "r3_pmc_summary.json")
# in-kernel clock and issue cost of the integer instructions the arithmetic is made of (tools/ubench_issue.hip, measured on MI355X:
# memtime / memrealtime under load; v_mul_lo/hi_u32, v_add3_u32, v_lshl_add_u64 4.4-4.6 cycles per wave-instruction per SIMD, v_mad_u64_u32 5.0-6.1)
VALU_CYCLES_PER_INST = 4.4
SUSTAINED_GHZ = 2.1
SIMDS = 1024


def a_block_bytes(n, L, K):
    """SURVEY 8(d) op-level algorithmic bytes of one transciphering"""
    P = 8 * n
    ks = 4 * L + 2 * L * K
    units = 518 * ks + 514 * 5 * L + 522 * 6 * L + 5 * 3 * L + 4 * L + 2 * 5 * L + 2 * 7 * L + 4 * (5 * L + 2 * L * K)
    return units * P


def ks_row_bytes_per_item(n, L, K):
    """bytes ks_row_kernel must move per ciphertext of a rotation step: reads the L*K forward
    intermediates, the accumulator of the plain product (L) and, for the c0 tiles of its grid, intermediate + S_0 + c0 +
    accumulator (4L); writes the inverse row passes (L + 2), S_0 (L), both accumulators (2L) and the next c0 (L).
    Key-switch keys and plaintext diagonals are shared by the whole batch (cache resident) and not counted."""
    return (L * K + L + 4 * L + (L + 2) + L + 2 * L + L) * 8 * n


def fc_row_bytes(n, L, K, n_inputs):
    """one FC row = multiply + relinearize + KS(n) rotations + (n-1) adds; KS from the NAF decompositions"""
    P = 8 * n

    def naf_terms(v):
        c = 0
        while v:
            if v & 1:
                z = 2 - (v & 3)
                v -= z
                c += 1
            v >>= 1
        return c
    ks = sum(1 if (i & (i - 1)) == 0 else naf_terms(i) for i in range(1, n_inputs))
    return (7 * L + (5 * L + 2 * L * K) + ks * (4 * L + 2 * L * K) + (n_inputs - 1) * 6 * L) * P, ks


def naf(v):
    """util::naf (seal/util/numth.h:22-42) of a negative step, terms in vector order"""
    out, bit, neg = [], 0, v < 0
    v = abs(v)
    while v:
        z = (2 - (v & 3)) if v & 1 else 0
        v = (v - z) >> 1
        if z:
            out.append((-z if neg else z) << bit)
        bit += 1
    return out


def fc_trie_key_switches(n, n_inputs):
    """key switches hhe_fc_row EXECUTES for one row with a default GaloisKeys object: the NAF term sequences of the steps -1..-(n_inputs-1)
    share prefixes and are evaluated as a trie (identical ciphertext words); terms equal to +-N/2 are skipped as SEAL does"""
    seen = set()
    for i in range(1, n_inputs):
        terms = [-i] if (i & (i - 1)) == 0 else [t for t in naf(-i) if abs(t) != n // 4 * 2]
        for k in range(1, len(terms) + 1):
            seen.add(tuple(terms[:k]))
    return len(seen)


def source_hash():
    """identifies the kernel sources a committed PMC profile belongs to"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, PKG, "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def synthetic_keys(rng, q, n):
    """uniform key-switch key words of SEAL's layout [L][2][K][N] (values irrelevant to throughput)"""
    K, L = len(q), len(q) - 1
    k = np.empty((L, 2, K, n), np.uint64)
    for j in range(K):
        k[:, :, j, :] = rng.integers(0, q[j], size=(L, 2, n), dtype=np.uint64)
    return k


def synthetic_ct(rng, q, n, count=None):
    L = len(q) - 1
    shape = (2, n) if count is None else (count, 2, n)
    return np.stack([rng.integers(0, q[j], size=shape, dtype=np.uint64) for j in range(L)], axis=-2)


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: this process never touches the GPU; it starts N fresh rank processes
    (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment), lets rank 0 print the one JSON line on the
    inherited stdout and fails if any rank fails."""
    import socket
    import subprocess
    if "HHE_BENCH_DEVICE" not in os.environ and "HHE_LIB" not in os.environ:
        import torch
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < n_ranks:
            print(f"bench.py: --gpus {n_ranks} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending and rc == 0:
            for p in list(pending):
                try:
                    r = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if r != 0:
                    rc = r
    finally:
        for p in procs:
            if p.poll() is None:
                if rc:
                    p.kill()  # a failed rank leaves the others stuck at the barrier
                p.wait()
    if rc:
        print(f"bench.py: a rank failed (exit code {rc})", file=sys.stderr)
    return 1 if rc else 0


class Device:
    """device-side plumbing of one rank: torch tensors, one stream, events.  `rehearsal` (an explicitly passed non-HIP
    library, HHE_LIB=tests/emu/libhhe_emu.so) keeps everything on the host so the multi-rank plumbing can be run on a CPU;
    its throughput is meaningless and the JSON line says so."""

    def __init__(self, torch, local_rank, rehearsal):
        self.torch, self.rehearsal = torch, rehearsal
        if rehearsal:
            self.dev, self.stream = "cpu", None
        else:
            assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
            self.dev = f"cuda:{local_rank}"
            torch.cuda.set_device(local_rank)
            self.stream = torch.cuda.Stream(device=self.dev)

    def to_dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(self.dev)

    def zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.int64, device=self.dev)

    def sync(self):
        if not self.rehearsal:
            self.torch.cuda.synchronize()

    def timed(self, fn):
        """(host seconds, device ms on the library's stream) of fn()"""
        self.sync()
        if self.rehearsal:
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            return dt, dt * 1e3
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(self.stream)
        fn()
        e1.record(self.stream)
        self.sync()
        return time.perf_counter() - t0, e0.elapsed_time(e1)


def make_context(api, lib, D, logn, q, t, local_rank):
    X = api.Context(logn, q, t, device=0 if D.rehearsal else local_rank, lib=lib)
    if D.stream is not None:
        X.set_stream(D.stream.cuda_stream)
    return X


def leg_config5(api, lib, D, local_rank, rng):
    """BASELINE config 5: N=2^16, 6x60-bit primes (L=5, K=6), t=8088322049: a 512-step rotate_rows(-1) chain over a batch
    (Evaluator::rotate_rows -> apply_galois + switch_key, seal/evaluator.h:955-1060), GB/s on SURVEY 8(d)'s per-op bytes."""
    logn, q, t = 16, Q_CONFIG5, T_CONFIG5
    n, K, L = 1 << logn, len(q), len(q) - 1
    B, steps = 32, 512
    X = make_context(api, lib, D, logn, q, t, local_rank)
    e = X.query("galois_elt", -1)
    X.set_galois_key(e, synthetic_keys(rng, q, n))
    a, b = D.to_dev(synthetic_ct(rng, q, n, B)), D.zeros((B, 2, L, n))
    X.rotate_rows(a, -1, b, B)  # warm-up (workspaces)
    X.sync()

    def chain():
        src, dst = a, b
        for _ in range(steps):
            X.rotate_rows(src, -1, dst, B)
            src, dst = dst, src
        X.sync()
    _, ms = D.timed(chain)
    per_rot = (4 * L + 2 * L * K) * 8 * n
    gbps = per_rot * B * steps / (ms * 1e-3) / 1e9
    X.close()
    return {"workload": f"BASELINE config 5: N=2^16, 6x60-bit primes (L=5,K=6), t={t}: {steps}-step rotate_rows(-1) chain, batch {B}",
            "rotations_per_s": B * steps / (ms * 1e-3), "ms_per_rotation_of_batch": ms / steps,
            "algorithmic_bytes_per_rotation": per_rot, "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
            "cpu_reference_ms_per_rotation": 58.1, "cpu_reference_source": "SURVEY 3.4 (SEAL 4.0.0, 1 core, measured by the survey)"}


class MnistFlow:
    """BASELINE config 3 on one rank: 784-word samples -> hhe_decompose (7 transcipherings + mask + flatten, CSP.cpp:235-283) and the
    784x10 FC (sealhelper.cpp:268-274, 379-392; CSP.cpp:306), with the key OBJECTS the CSP names at each call as key sets: the
    PASTA_SEAL's RelinKeys / GaloisKeys (analyst's, CSP.cpp:238-242), the flatten GaloisKeys (csp_gk, hhe_pktnn_examples.cpp:601-615),
    the CSP's RelinKeys for the FC (CSP.cpp:306) and the analyst's default GaloisKeys for the slot sums (CSP.cpp:312-316).
    Synthetic keys / data (uniform words of SEAL's shapes): throughput only; parity for this flow is tests/test_gpu_parity.py."""

    def __init__(self, api, lib, D, local_rank, logn, q, t, nin=784, neurons=10):
        self.D, self.logn, self.q, self.t, self.nin, self.neurons = D, logn, q, t, nin, neurons
        self.n, self.K, self.L = 1 << logn, len(q), len(q) - 1
        n = self.n
        self.nblocks = (nin + 127) // 128
        self.X = X = make_context(api, lib, D, logn, q, t, local_rank)
        rng = np.random.default_rng(4321)   # every rank holds the same (replicated) keys and weights
        self.pasta, self.csp_gk, self.analyst_gk, self.csp_rk = X.keyset(), X.keyset(), X.keyset(), X.keyset()
        self.pasta.set_relin(synthetic_keys(rng, q, n))
        self.csp_rk.set_relin(synthetic_keys(rng, q, n))
        pasta_steps = [0, -1] + ([128] if n // 2 != 128 else [])
        for s_ in pasta_steps:
            self.pasta.set_galois(X.query("galois_elt", s_), synthetic_keys(rng, q, n))
        for s_ in pasta_steps + [-128 * i for i in range(1, self.nblocks)]:
            self.csp_gk.set_galois(X.query("galois_elt", s_), synthetic_keys(rng, q, n))
        elts, g, gi = {2 * n - 1}, 3, pow(3, -1, 2 * n)
        for _ in range(logn - 1):  # GaloisKeys created without arguments: 3^(2^k), their inverses and the column swap (Analyst.cpp:62-65)
            elts.add(g)
            elts.add(gi)
            g, gi = g * g % (2 * n), gi * gi % (2 * n)
        for e in sorted(elts):
            self.analyst_gk.set_galois(e, synthetic_keys(rng, q, n))
        self.enc_key = D.to_dev(synthetic_ct(rng, q, n))
        self.w = D.to_dev(synthetic_ct(rng, q, n, neurons))

    def records(self, lo, hi):
        """symmetric-ciphertext words of samples [lo, hi): seeded per sample, so a rank's share does not depend on the rank count"""
        return np.stack([np.random.default_rng(1000 + s_).integers(0, self.t, self.nin, dtype=np.uint64) for s_ in range(lo, hi)]) if hi > lo \
            else np.zeros((0, self.nin), np.uint64)

    def run(self, recs, warm=False):
        """decompose + FC of `recs`; returns (decompose ms, FC ms) on the library's stream"""
        D, X, S = self.D, self.X, len(recs)
        flat = D.zeros((S, 2, self.L, self.n))
        _, dec_ms = D.timed(lambda: X.decompose(self.enc_key, recs, flat, rk=self.pasta, gk=self.pasta, flatten_gk=self.csp_gk))
        vi = flat.repeat_interleave(self.neurons, dim=0).contiguous() if hasattr(flat, "repeat_interleave") else \
            np.repeat(flat, self.neurons, axis=0)  # item = (sample, neuron), neuron = item % neurons
        out = D.zeros(tuple(vi.shape))
        _, fc_ms = D.timed(lambda: X.fc_row(vi, self.w, self.neurons, self.nin, out, S * self.neurons, rk=self.csp_rk, gk=self.analyst_gk))
        return dec_ms, fc_ms

    def rooflines(self, S, dec_ms, fc_ms):
        n, L, K = self.n, self.L, self.K
        P = 8 * n
        fc_bytes, ks_ref = fc_row_bytes(n, L, K, self.nin)
        ks_exec = fc_trie_key_switches(n, self.nin)
        fc_exec_bytes = (7 * L + (5 * L + 2 * L * K) + ks_exec * (4 * L + 2 * L * K) + (self.nin - 1) * 6 * L) * P
        rows = S * self.neurons
        fc_gbps, fc_exec_gbps = fc_bytes * rows / (fc_ms * 1e-3) / 1e9, fc_exec_bytes * rows / (fc_ms * 1e-3) / 1e9
        nb = self.nblocks
        dec_bytes = nb * a_block_bytes(n, L, K) + (nb - 1) * (4 * L + 2 * L * K + 6 * L) * P + 5 * L * P  # blocks + flatten + mask
        dec_gbps = dec_bytes * S / (dec_ms * 1e-3) / 1e9
        return {"fc_roofline": {"bound": "hbm", "algorithmic_bytes_per_row": fc_exec_bytes, "key_switches_per_row_executed": ks_exec,
                                "rows": rows, "achieved": fc_exec_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fc_exec_gbps / HBM_PEAK_GBPS,
                                "reference_op_count": {"key_switches_per_row": ks_ref, "algorithmic_bytes_per_row": fc_bytes, "equivalent_GBps": fc_gbps,
                                                       "ratio_to_peak": fc_gbps / HBM_PEAK_GBPS,
                                                       "note": "SURVEY 8(d)'s FC formula on the reference's own NAF key switches per row; NOT a roofline "
                                                               "fraction (it may exceed 1): the rotation trie evaluates fewer key switches with identical ciphertext words"},
                                "note": "`achieved` / `frac` price the key switches the rotation trie executes at SURVEY 8(d)'s bytes per key switch (an upper "
                                        "bound on the bytes the path needs: the data-limb inner products of its leaves are merged per Galois element)"},
                "decompose_roofline": {"algorithmic_bytes_per_sample": dec_bytes, "achieved": dec_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": dec_gbps / HBM_PEAK_GBPS}}

    def close(self):
        for ks in (self.pasta, self.csp_gk, self.analyst_gk, self.csp_rk):
            ks.close()
        self.X.close()


def leg_mnist(api, lib, D, local_rank):
    """BASELINE config 3 shape on one GPU (extras of the default line)"""
    F = MnistFlow(api, lib, D, local_rank, 15, Q_CONFIG2, T_PLAIN)
    S = 16
    F.run(F.records(0, 1))  # warm-up: public tables of the 7 block counters, workspaces, correction tables of the Galois keys
    dec_ms, fc_ms = F.run(F.records(0, S))
    fallbacks = F.X.query("fc_fallbacks")
    out = {"workload": f"BASELINE config 3 shape: {S} samples of 784 words, N=2^15, 4x60-bit, t=65537, synthetic keys; decompose + 784x10 FC",
           "samples_per_s": S / ((dec_ms + fc_ms) * 1e-3), "decompose_ms_per_sample": dec_ms / S, "fc_ms_per_sample": fc_ms / S,
           "transcipherings_per_s_in_decompose": 7 * S / (dec_ms * 1e-3), "fc_shared_digit_fallbacks": int(fallbacks)}
    out.update(F.rooflines(S, dec_ms, fc_ms))
    F.close()
    return out


def leg_record_latency(api, lib, D, local_rank):
    """The drop-in's call shape: BaseCSP::decompose calls decomposition once per RECORD (CSP.cpp:247-252) -- 3 blocks for the
    gRPC flow's 300 words (CSPRPC.cpp:196), 7 for an MNIST sample.  ms per hhe_decompose call of ONE record after warm-up."""
    out = {}
    for nin in (784, 300):
        F = MnistFlow(api, lib, D, local_rank, 15, Q_CONFIG2, T_PLAIN, nin=nin)
        rec = F.records(0, 1)
        flat = D.zeros((1, 2, F.L, F.n))
        call = lambda: F.X.decompose(F.enc_key, rec, flat, rk=F.pasta, gk=F.pasta, flatten_gk=F.csp_gk)  # noqa: E731
        call()
        call()
        reps = 5
        host_s, ms = D.timed(lambda: [call() for _ in range(reps)])
        out[f"{nin}_words"] = {"blocks": F.nblocks, "ms_per_call": ms / reps, "host_ms_per_call": host_s * 1e3 / reps,
                               "transcipherings_per_s": F.nblocks * reps / (ms * 1e-3)}
        F.close()
    out["workload"] = "one record per hhe_decompose call (N=2^15, 4x60-bit, t=65537): transcipher its blocks + mask + flatten, key sets resident"
    return out


def leg_reference_defaults(api, lib, D, local_rank, rng):
    """the only decrypt-correct configuration the reference ships (configs/config.cpp:19-20): N=2^14, BFVDefault 9 primes, t=65537"""
    logn, t = 14, T_PLAIN
    q = api.bfv_default_coeff_modulus(1 << logn, lib)
    n, K, L = 1 << logn, len(q), len(q) - 1
    B, steps = 256, 2
    X = make_context(api, lib, D, logn, q, t, local_rank)
    for step in (-1, 128, 0):
        X.set_galois_key(X.query("galois_elt", step), synthetic_keys(rng, q, n))
    X.set_relin_key(synthetic_keys(rng, q, n))
    d_key = D.to_dev(synthetic_ct(rng, q, n))
    cw = rng.integers(0, t, size=(B, 128), dtype=np.uint64)
    out = D.zeros((B, 2, L, n))
    X.reserve(B)
    run = lambda: X.transcipher(d_key, cw, np.full(B, 128, np.uint32), np.zeros(B, np.uint64), out)  # noqa: E731
    run()
    _, ms = D.timed(lambda: [run() for _ in range(steps)])
    X.close()
    A = a_block_bytes(n, L, K)
    per_s = B * steps / (ms * 1e-3)
    return {"workload": f"reference defaults: N=2^14, BFVDefault 9 primes (L=8,K=9), t=65537, batch-{B} independent 128-word blocks, block counter 0",
            "transcipherings_per_s": per_s, "ms_per_step": ms / steps, "algorithmic_bytes_per_unit": A,
            "path_achieved_GBps": A * per_s / 1e9, "path_frac_of_8TBps": A * per_s / 1e9 / HBM_PEAK_GBPS}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="transcipherings per rank per step (default 256); mnist-e2e: samples per rank per step (default 16)")
    ap.add_argument("--workload", default="config2", choices=["config2", "mnist", "mnist-e2e"],
                    help="config2: all blocks use counter 0 (BASELINE metric); mnist: 784-word samples = blocks 0..6 (last ragged), transciphering only; "
                         "mnist-e2e: BASELINE config 3 -- every rank runs hhe_decompose + the FC rows on its contiguous sample range, --batch = samples per rank")
    ap.add_argument("--record-words", type=int, default=784, help="mnist-e2e: words per sample")
    ap.add_argument("--neurons", type=int, default=10, help="mnist-e2e: FC rows per sample")
    ap.add_argument("--params", default="config2", choices=["config2", "default16384", "tiny"],
                    help="config2: N=2^15, 4x60-bit (BASELINE metric); default16384: the reference's defaults N=2^14, BFVDefault 9 primes; "
                         "tiny: N=2^10 (plumbing rehearsals only)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-blocks-per-thread", type=int, default=3)
    ap.add_argument("--extras", type=int, default=1, help="config-5 rotation chain and MNIST-shaped legs (N=1 only)")
    ap.add_argument("--kernel-timing", type=int, default=1, help="extra step with HIP events around every launch of the dominant kernel (roofline block)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    import torch
    sh = importlib.import_module(PKG + ".sharding")
    api = importlib.import_module(PKG + ".api")
    rank, world, local_rank = sh.rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    lib = api.load_library()  # the in-tree gfx950 library, or the one HHE_LIB names explicitly; raises if missing
    backend = lib.hhe_backend().decode()
    rehearsal = backend != "hip-gfx950"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL; used for the barrier / time reduction only (HHE_BENCH_BACKEND=gloo: rehearsal of the N>1 path on one GPU)
        sh.init_process_group("gloo" if rehearsal else os.environ.get("HHE_BENCH_BACKEND", "nccl"))
    if "HHE_BENCH_DEVICE" in os.environ:  # rehearsal only: several ranks on one card
        local_rank = int(os.environ["HHE_BENCH_DEVICE"])
    D = Device(torch, local_rank, rehearsal)

    logn, q, t = 15, Q_CONFIG2, T_PLAIN
    if args.params == "default16384":
        logn = 14
        q = api.bfv_default_coeff_modulus(1 << logn, lib)
    elif args.params == "tiny":
        logn, q = 10, [1125899906738177, 1125899906820097, 1125899906826241]  # CoeffModulus::Create(1024, {50,50,50})
    n, K, L = 1 << logn, len(q), len(q) - 1
    if args.workload == "mnist-e2e":
        return main_mnist_e2e(args, api, lib, sh, D, rank, world, local_rank, rehearsal, backend, logn, q, t)
    B = args.batch or 256
    X = make_context(api, lib, D, logn, q, t, local_rank)
    rng = np.random.default_rng(1234)
    # synthetic key material + encrypted PASTA key (uniform words; same shapes as SEAL's objects)
    rk = synthetic_keys(rng, q, n)
    gks = {}
    for step in (-1, 128, 0):
        e = X.query("galois_elt", step)
        gks[e] = synthetic_keys(rng, q, n)
        X.set_galois_key(e, gks[e])
    X.set_relin_key(rk)
    enc_key = synthetic_ct(rng, q, n)
    d_key = D.to_dev(enc_key)
    # B independent 128-word symmetric ciphertext blocks (uniform words < t), block counter 0 (config 2)
    cw = rng.integers(0, t, size=(B, 128), dtype=np.uint64)
    ncw = np.full(B, 128, np.uint32)
    bidx = np.zeros(B, np.uint64)
    if args.workload == "mnist":  # BASELINE config 3 shape: samples of 784 words -> 7 blocks, sharded by sample
        bidx = (np.arange(B) % 7).astype(np.uint64)
        ncw = np.where(bidx == 6, 16, 128).astype(np.uint32)
    out = D.zeros((B, 2, L, n))
    X.reserve(B)

    def step():
        X.transcipher(d_key, cw, ncw, bidx, out)

    for _ in range(args.warmup):
        step()
    D.sync()
    sh.barrier()
    elapsed, dev_ms = D.timed(lambda: [step() for _ in range(args.steps)])
    sh.barrier()
    red_dev = D.dev if (world > 1 and not rehearsal and os.environ.get("HHE_BENCH_BACKEND", "nccl") == "nccl") else "cpu"
    el_max, units = sh.reduce_max_sum(elapsed, B * args.steps, device=red_dev)
    value = units / el_max

    res = None
    if rank == 0:
        A = a_block_bytes(n, L, K)
        path_ms = dev_ms / args.steps
        path_gbps = A * B / (path_ms * 1e-3) / 1e9
        label = {"config2": "PASTA-3 transcipherings/sec (N=2^15, 4 RNS limbs)",
                 "default16384": "PASTA-3 transcipherings/sec (N=2^14, BFVDefault 9 primes)",
                 "tiny": "PASTA-3 transcipherings/sec (N=2^10 plumbing rehearsal)"}[args.params]
        wl = {"config2": "BASELINE config 2: N=2^15, coeff_modulus 4x60-bit (L=3,K=4), t=65537",
              "default16384": "reference defaults: N=2^14, BFVDefault 9 primes (L=8,K=9), t=65537",
              "tiny": "N=2^10, 3x50-bit primes, t=65537"}[args.params]
        wl += (f", batch-{B} independent 128-word PASTA-3 blocks per GPU, block counter 0" if args.workload == "config2" else
               f", batch-{B} blocks per GPU = 784-word samples x 7 block counters (last block 16 words)")
        res = {"metric": label, "value": value, "unit": "transcipherings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": el_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u64", "data": "synthetic", "backend": backend,
               "config": {"workload": wl, "batch_per_gpu": B, "sharding": f"{world} rank(s), independent items, no collective"}}
        # ---- roofline: the dominant kernel, timed live on the stream it is launched on
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
        if logn >= 12 and not rehearsal and args.kernel_timing:
            # one more step of the same workload, identical configuration, with every launch of the kernel bracketed by HIP
            # events on the stream it is launched on (the library's internal streams): the same launches rocprofv3's
            # kernel trace of this command averages (profiles/)
            X.profile(True)
            step()
            kname, launches, total_ms, items = X.profile_read()
            X.profile(False)
            roof["kernel"] = kname
            if launches == 0:
                roof["note"] = "the fused key-switch row kernel did not run in this configuration (HHE_MATMUL=0 or moduli without the pseudo-Mersenne form): nothing to time"
            else:
                avg_us = total_ms * 1e3 / launches
                per_launch = ks_row_bytes_per_item(n, L, K) * items / launches
                ach = per_launch / (avg_us * 1e-6) / 1e9
                roof.update({"achieved": ach, "frac": ach / HBM_PEAK_GBPS, "launches_timed": launches,
                             "avg_launch_us": avg_us, "ciphertexts_per_launch": items / launches,
                             "algorithmic_bytes_per_launch": per_launch,
                             "algorithmic_bytes_definition": "per ciphertext of a rotation step: (L*K + 5L) P read + (5L + 2) P written, P = 8N (DESIGN.md section 4)",
                             "hbm_min_us": per_launch / (HBM_PEAK_GBPS * 1e3),
                             "kernel_time_over_path_time": total_ms / path_ms,
                             "note": "the chunks of a batch run one after the other on one internal stream (HHE_STREAMS=1, the default), so a launch's duration is its own; with HHE_STREAMS=2 launches of the two streams overlap and their durations stretch"})
                # PMC-derived figures of the same kernel come from a committed rocprofv3 run; they are only reported while the
                # kernel sources are the ones that run was made with
                if os.path.exists(PMC_PROFILE) and args.params == "config2":
                    pm = json.load(open(PMC_PROFILE))
                    stale = pm.get("source_hash") != source_hash()
                    roof["traffic_source"] = {"file": "profiles/" + os.path.basename(PMC_PROFILE), "source_hash": pm.get("source_hash"), "stale": stale,
                                              "how": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and --pmc WRITE_SIZE, separate passes (tools/pmc_passes.sh)"}
                    if not stale:
                        kr = pm["kernels"].get("ks_row_kernel", {})
                        scale = (items / launches) / pm.get("batch", items / launches)   # the PMC passes run one chunk of `batch` ciphertexts per launch
                        if kr.get("traffic_bytes_per_launch"):
                            roof["traffic"] = kr["traffic_bytes_per_launch"] * scale
                            roof["traffic_GBps_at_measured_duration"] = roof["traffic"] / (avg_us * 1e-6) / 1e9
                        if kr.get("valu_wave_instructions_per_launch"):
                            wi = kr["valu_wave_instructions_per_launch"] * scale
                            valu_min_us = wi * VALU_CYCLES_PER_INST / SIMDS / (SUSTAINED_GHZ * 1e3)
                            roof["valu_ceiling"] = {"wave_instructions_per_launch": wi, "issue_cycles_per_wave_instruction": VALU_CYCLES_PER_INST, "simds": SIMDS,
                                                    "clock_ghz": SUSTAINED_GHZ, "min_us": valu_min_us, "valu_busy_at_measured_duration": valu_min_us / avg_us,
                                                    "note": "integer instructions issue at one wave-instruction per ~4.4 cycles per SIMD at the ~2.1 GHz the chip holds under this load (tools/ubench_issue.hip, in-kernel clock): the kernel cannot run faster than min_us"}
                            # the nearer of the two limits names the bound; `frac` stays the HBM fraction the contract asks for
                            roof["bound"] = "hbm" if roof["hbm_min_us"] >= valu_min_us else "valu-issue"
                            roof["bound_detail"] = {"hbm_min_us": roof["hbm_min_us"], "valu_issue_min_us": valu_min_us, "measured_us": avg_us,
                                                    "note": "neither limit is reached: with 4 waves per SIMD (128 VGPRs) the kernel waits on dependent global -> LDS round trips (DESIGN.md section 5, stamps timeline)"}
                        res["path_traffic_bytes_per_transciphering"] = pm.get("traffic_bytes_per_transciphering")
                        res["path_valu_wave_instructions_per_transciphering"] = pm.get("valu_wave_instructions_per_transciphering")
        roof["path"] = {"kernel": "hhe_pasta3_transcipher (whole path; SURVEY 8d op-level model A_block)", "algorithmic_bytes_per_unit": A,
                        "units_per_launch": B, "launch_ms": path_ms, "achieved": path_gbps, "frac": path_gbps / HBM_PEAK_GBPS}
        res["roofline"] = roof
        # ---- CPU baseline: the oracle port on the host cores (N = 1 only)
        res["cpu_baseline"] = None
        if world == 1 and args.cpu_baseline and not rehearsal:
            import oracle as orc  # CPU baseline leg only
            O = orc.Oracle(logn, q, t)
            threads = min(16, len(os.sched_getaffinity(0)))  # the GPU box grants 16 cores per GPU
            nb = threads * args.cpu_blocks_per_thread
            elts = sorted(gks)
            gk = orc.GaloisKeys(elts, np.stack([gks[e] for e in elts]))
            c0 = time.perf_counter()
            ref1 = O.transcipher_batch(enc_key, rk, gk, cw[:1], ncw[:1], bidx[:1], threads=1)  # warm-up item = the 1-core figure
            one_s = time.perf_counter() - c0
            idx = np.arange(nb) % B
            c0 = time.perf_counter()
            O.transcipher_batch(enc_key, rk, gk, cw[idx], ncw[idx], bidx[idx], threads=threads)
            cpu_s = time.perf_counter() - c0
            got = out[:1].cpu().numpy().view(np.uint64)
            res["cpu_baseline"] = {"value": nb / cpu_s, "unit": "transcipherings/s", "cores": threads, "kind": "port",
                                   "cpu_model": cpu_model(), "one_core_value": 1.0 / one_s,
                                   "sample": f"{nb} blocks of the same workload ({args.cpu_blocks_per_thread} per thread, OpenMP over blocks) in {cpu_s:.1f} s after a 1-block warm-up on one core ({one_s:.1f} s)",
                                   "matches_gpu_item0": bool((got[0] == ref1[0]).all())}
        # ---- BASELINE configs 5 and 3 (N = 1 only)
        if world == 1 and args.extras and args.params == "config2" and not rehearsal:
            X.close()
            X = None
            del out
            torch.cuda.empty_cache()
            res["extras"] = {"config5_rotate_chain": leg_config5(api, lib, D, local_rank, rng),
                             "mnist_1fc": leg_mnist(api, lib, D, local_rank),
                             "reference_defaults": leg_reference_defaults(api, lib, D, local_rank, rng),
                             "per_record_latency": leg_record_latency(api, lib, D, local_rank)}
        print(json.dumps(res), flush=True)
    if X is not None:
        X.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main_mnist_e2e(args, api, lib, sh, D, rank, world, local_rank, rehearsal, backend, logn, q, t):
    """BASELINE config 3: samples shard contiguously over the ranks (sharding.shard_samples), every rank runs hhe_decompose + the FC rows
    on its range with replicated keys / weights, no data-path collective; value = samples of all ranks / max-over-ranks time."""
    per_rank = args.batch or 16
    total = per_rank * world   # weak scaling: the per-GPU share is fixed
    lo, hi = sh.shard_samples(total, rank, world)
    F = MnistFlow(api, lib, D, local_rank, logn, q, t, nin=args.record_words, neurons=args.neurons)
    recs = F.records(lo, hi)
    for _ in range(max(1, args.warmup)):   # the first call builds the public tables of the block counters and the workspaces
        F.run(recs[:1])
    D.sync()
    sh.barrier()
    dec_ms = fc_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d_, f_ = F.run(recs)
        dec_ms += d_
        fc_ms += f_
    D.sync()
    elapsed = time.perf_counter() - t0
    sh.barrier()
    red_dev = D.dev if (world > 1 and not rehearsal and os.environ.get("HHE_BENCH_BACKEND", "nccl") == "nccl") else "cpu"
    el_max, units = sh.reduce_max_sum(elapsed, (hi - lo) * args.steps, device=red_dev)
    if rank == 0:
        S = hi - lo
        wl = {"config2": "N=2^15, coeff_modulus 4x60-bit (L=3,K=4), t=65537", "default16384": "N=2^14, BFVDefault 9 primes (L=8,K=9), t=65537",
              "tiny": "N=2^10, 3x50-bit primes, t=65537 (plumbing rehearsal)"}[args.params]
        res = {"metric": f"MNIST-shaped samples/sec: PASTA-3 transcipher ({F.nblocks} blocks) + mask + flatten + {args.record_words}x{args.neurons} encrypted FC",
               "value": units / el_max, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": el_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
               "data": "synthetic", "backend": backend,
               "config": {"workload": f"BASELINE config 3: {wl}; {per_rank} samples of {args.record_words} words per GPU and step, {args.neurons} FC rows per sample, synthetic keys and data",
                          "batch_per_gpu": per_rank, "sharding": f"{world} rank(s), contiguous sample ranges, replicated keys and weights, no collective"},
               "rank0": {"samples": S, "decompose_ms_per_sample": dec_ms / args.steps / max(S, 1), "fc_ms_per_sample": fc_ms / args.steps / max(S, 1),
                         "transcipherings_per_s_in_decompose": F.nblocks * S * args.steps / (dec_ms * 1e-3) if dec_ms else None}}
        if S:
            rl = F.rooflines(S, dec_ms / args.steps, fc_ms / args.steps)
            res["roofline"] = dict(rl["fc_roofline"], kernel="hhe_fc_row (the larger share of a sample; SURVEY 8d op-level model)", traffic=None,
                                   decompose=rl["decompose_roofline"])
        res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    F.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
