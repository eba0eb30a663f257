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
extras    = BASELINE configs 5 and 3 (N=1 only): a 512-step rotate_rows chain at N=2^16 / 6 primes, and MNIST-shaped
            samples through hhe_decompose + the 784x10 FC.
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
PMC_PROFILE = os.path.join(ROOT, "profiles", "r2_pmc_summary.json")


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


def leg_mnist(api, lib, D, local_rank, rng):
    """BASELINE config 3 shape on one GPU: 784-word samples -> hhe_decompose (7 transcipherings + mask + flatten,
    CSP.cpp:235-283) and the 784x10 FC (sealhelper.cpp:268-274,379-392; CSP.cpp:306).  Synthetic keys / data at the metric's
    parameters (the noise budget is 0 there, SURVEY 3.4: throughput only; parity for this flow is tests/test_gpu_parity.py)."""
    logn, q, t = 15, Q_CONFIG2, T_PLAIN
    n, K, L = 1 << logn, len(q), len(q) - 1
    S, OUT, NIN = 16, 10, 784
    X = make_context(api, lib, D, logn, q, t, local_rank)
    X.set_relin_key(synthetic_keys(rng, q, n))
    X.set_relin_key_slot(1, synthetic_keys(rng, q, n))
    elts = {X.query("galois_elt", s) for s in [0, -1, 128] + [-128 * i for i in range(1, 7)]}
    g, gi = 3, pow(3, -1, 2 * n)
    for _ in range(logn - 1):  # GaloisKeys created without arguments: 3^(2^k) and their inverses
        elts.add(g)
        elts.add(gi)
        g, gi = g * g % (2 * n), gi * gi % (2 * n)
    for e in sorted(elts):
        X.set_galois_key(e, synthetic_keys(rng, q, n))
    enc_key = D.to_dev(synthetic_ct(rng, q, n))
    recs = rng.integers(0, t, size=(S, NIN), dtype=np.uint64)
    flat = D.zeros((S, 2, L, n))
    w = D.to_dev(synthetic_ct(rng, q, n, OUT))
    X.decompose(enc_key, recs[:1], flat[:1])  # warm-up: public tables of the 7 block counters, workspaces
    _, dec_ms = D.timed(lambda: X.decompose(enc_key, recs, flat))
    vi = flat.repeat_interleave(OUT, dim=0).contiguous()  # item = (sample, neuron), neuron = item % 10
    out = D.zeros(tuple(vi.shape))
    X.fc_row(vi[:OUT], w, OUT, NIN, out[:OUT], OUT, relin_slot=1)  # warm-up: correction tables of the Galois keys
    _, fc_ms = D.timed(lambda: X.fc_row(vi, w, OUT, NIN, out, S * OUT, relin_slot=1))
    fallbacks = X.query("fc_fallbacks")
    X.close()
    fc_bytes, ks = fc_row_bytes(n, L, K, NIN)
    fc_gbps = fc_bytes * S * OUT / (fc_ms * 1e-3) / 1e9
    dec_bytes = 7 * a_block_bytes(n, L, K) + 6 * (4 * L + 2 * L * K + 6 * L) * 8 * n + 5 * L * 8 * n  # 7 blocks + flatten + mask
    dec_gbps = dec_bytes * S / (dec_ms * 1e-3) / 1e9
    return {"workload": f"BASELINE config 3 shape: {S} samples of 784 words, N=2^15, 4x60-bit, t=65537, synthetic keys; decompose + 784x10 FC",
            "samples_per_s": S / ((dec_ms + fc_ms) * 1e-3), "decompose_ms_per_sample": dec_ms / S, "fc_ms_per_sample": fc_ms / S,
            "transcipherings_per_s_in_decompose": 7 * S / (dec_ms * 1e-3),
            "fc_roofline": {"bound": "hbm", "algorithmic_bytes_per_row": fc_bytes, "key_switches_per_row_in_the_model": ks,
                            "rows": S * OUT, "achieved": fc_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fc_gbps / HBM_PEAK_GBPS,
                            "note": "SURVEY 8(d) FC formula counts the reference's 2875 key switches per row; the rotation trie evaluates 1054"},
            "decompose_roofline": {"algorithmic_bytes_per_sample": dec_bytes, "achieved": dec_gbps, "frac": dec_gbps / HBM_PEAK_GBPS},
            "fc_shared_digit_fallbacks": int(fallbacks)}


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
    ap.add_argument("--batch", type=int, default=256, help="transcipherings per rank per step")
    ap.add_argument("--workload", default="config2", choices=["config2", "mnist"],
                    help="config2: all blocks use counter 0 (BASELINE metric); mnist: 784-word samples = blocks 0..6 (last ragged)")
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
    B = args.batch
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
            nb = B
            X.profile(True)
            step()
            kname, launches, total_ms, items = X.profile_read()
            X.profile(False)
            avg_us = total_ms * 1e3 / launches
            per_launch = ks_row_bytes_per_item(n, L, K) * items / launches
            ach = per_launch / (avg_us * 1e-6) / 1e9
            roof.update({"achieved": ach, "frac": ach / HBM_PEAK_GBPS, "kernel": kname, "launches_timed": launches,
                         "avg_launch_us": avg_us, "ciphertexts_per_launch": items / launches,
                         "algorithmic_bytes_per_launch": per_launch,
                         "algorithmic_bytes_definition": "per ciphertext of a rotation step: (L*K + 5L) P read + (5L + 2) P written, P = 8N (DESIGN.md section 4)",
                         "kernel_time_over_path_time": total_ms / path_ms,
                         "note": "the chunks of a batch run one after the other on one internal stream (HHE_STREAMS=1, the default), so a launch's duration is its own; with HHE_STREAMS=2 launches of the two streams overlap and their durations stretch"})
            # PMC-derived figures of the same kernel come from a committed rocprofv3 run; they are only reported while the
            # kernel sources are the ones that run was made with
            if os.path.exists(PMC_PROFILE) and args.params == "config2":
                pm = json.load(open(PMC_PROFILE))
                stale = pm.get("source_hash") != source_hash()
                roof["traffic_source"] = {"file": "profiles/r2_pmc_summary.json", "source_hash": pm.get("source_hash"), "stale": stale,
                                          "how": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and --pmc WRITE_SIZE, separate passes (tools/pmc_passes.sh)"}
                if not stale:
                    kr = pm["kernels"].get("ks_row_kernel", {})
                    roof["traffic"] = kr.get("traffic_bytes_per_launch")
                    roof["traffic_GBps_at_measured_duration"] = (kr.get("traffic_bytes_per_launch", 0) / (avg_us * 1e-6) / 1e9) if kr else None
                    if kr.get("valu_wave_instructions_per_launch"):
                        wi = kr["valu_wave_instructions_per_launch"]
                        roof["valu_ceiling"] = {"wave_instructions_per_launch": wi, "issue_cycles_per_wave_instruction": 4, "simds": 1024,
                                                "clock_ghz": 2.4, "min_us": wi * 4 / 1024 / 2.4e3,
                                                "note": "64-bit modular arithmetic issues one wave instruction per 4 cycles per SIMD (tools/ubench_intmul.hip, tools/ubench_bfly.hip); the kernel cannot run faster than this at 2.4 GHz"}
                    res["path_traffic_bytes_per_transciphering"] = pm.get("traffic_bytes_per_transciphering")
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
                             "mnist_1fc": leg_mnist(api, lib, D, local_rank, rng)}
        print(json.dumps(res), flush=True)
    if X is not None:
        X.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
