#!/usr/bin/env python3
"""bench.py -- PASTA-3 transcipherings/sec on MI355X (BASELINE.json metric).

step      = one pass of the hot path (hhe_pasta3_transcipher) over one batch of B independent
            128-word blocks per rank (BASELINE config 2: N=2^15, 4x60-bit primes, t=65537, B=256).
value     = transcipherings all ranks completed / max-over-ranks time (inputs resident in HBM;
            only the 1 KiB/block of symmetric ciphertext words crosses PCIe inside the timed region).
roofline  = SURVEY 8(d): algorithmic bytes of one transciphering (A_block) x blocks per launch of the
            path / its HIP-event time, against 8 TB/s; plus the dominant kernel measured by itself.
cpu_baseline = the CPU oracle (a C port of the reference schedule) on the host cores, rank 0, N=1 only.
"""
import argparse
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
T_PLAIN = 65537
LOGN = 15


def a_block_bytes(n, L, K):
    """SURVEY 8(d) op-level algorithmic bytes of one transciphering"""
    P = 8 * n
    ks = 4 * L + 2 * L * K
    units = 518 * ks + 514 * 5 * L + 522 * 6 * L + 5 * 3 * L + 4 * L + 2 * 5 * L + 2 * 7 * L + 4 * (5 * L + 2 * L * K)
    return units * P


def synthetic_keys(rng, q, n):
    """uniform key-switch key words of SEAL's layout [L][2][K][N] (values irrelevant to throughput)"""
    K, L = len(q), len(q) - 1
    k = np.empty((L, 2, K, n), np.uint64)
    for j in range(K):
        k[:, :, j, :] = rng.integers(0, q[j], size=(L, 2, n), dtype=np.uint64)
    return k


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="transcipherings per rank per step")
    ap.add_argument("--workload", default="config2", choices=["config2", "mnist"],
                    help="config2: all blocks use counter 0 (BASELINE metric); mnist: 784-word samples = blocks 0..6 (last ragged)")
    ap.add_argument("--params", default="config2", choices=["config2", "default16384"],
                    help="config2: N=2^15, 4x60-bit (BASELINE metric); default16384: the reference's defaults N=2^14, BFVDefault 9 primes")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-blocks-per-thread", type=int, default=2)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    import torch
    sh = importlib.import_module(PKG + ".sharding")
    api = importlib.import_module(PKG + ".api")
    rank, world, local_rank = sh.rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL; used for the barrier / time reduction only (HHE_BENCH_BACKEND=gloo: rehearsal of the N>1 path on one GPU)
        sh.init_process_group(os.environ.get("HHE_BENCH_BACKEND", "nccl"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    if "HHE_BENCH_DEVICE" in os.environ:  # rehearsal only: several ranks on one card
        local_rank = int(os.environ["HHE_BENCH_DEVICE"])
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    lib = api.load_library()

    logn = LOGN
    n, q, t = 1 << LOGN, Q_CONFIG2, T_PLAIN
    if args.params == "default16384":
        logn, n = 14, 1 << 14
        q = api.bfv_default_coeff_modulus(n, lib)
    K, L = len(q), len(q) - 1
    B = args.batch
    X = api.Context(logn, q, t, device=local_rank, lib=lib)
    stream = torch.cuda.Stream(device=dev)
    X.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(1234)
    # synthetic key material + encrypted PASTA key (uniform words; same shapes as SEAL's objects)
    rk = synthetic_keys(rng, q, n)
    gks = {}
    for step in (-1, 128, 0):
        e = X.query("galois_elt", step)
        gks[e] = synthetic_keys(rng, q, n)
        X.set_galois_key(e, gks[e])
    X.set_relin_key(rk)
    enc_key = np.stack([rng.integers(0, q[j], size=(2, n), dtype=np.uint64) for j in range(L)], axis=1)
    d_key = torch.from_numpy(enc_key.view(np.int64)).to(dev)
    # B independent 128-word symmetric ciphertext blocks (uniform words < t), block counter 0 (config 2)
    cw = rng.integers(0, t, size=(B, 128), dtype=np.uint64)
    ncw = np.full(B, 128, np.uint32)
    bidx = np.zeros(B, np.uint64)
    if args.workload == "mnist":  # BASELINE config 3 shape: samples of 784 words -> 7 blocks, sharded by sample
        bidx = (np.arange(B) % 7).astype(np.uint64)
        ncw = np.where(bidx == 6, 16, 128).astype(np.uint32)
    out = torch.zeros((B, 2, L, n), dtype=torch.int64, device=dev)
    X.reserve(B)

    def step():
        X.transcipher(d_key, cw, ncw, bidx, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sh.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    sh.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    red_dev = dev if (world > 1 and os.environ.get("HHE_BENCH_BACKEND", "nccl") == "nccl") else "cpu"
    el_max, units = sh.reduce_max_sum(elapsed, B * args.steps, device=red_dev)
    value = units / el_max

    # dominant kernel (ntt_pass_kernel) measured by itself with HIP events on the library's stream, in the launch mix
    # of one rotation step of the pipeline at its chunk size: forward over the 32x12 digit polys together with the 32x3
    # c0 limbs of the previous step (one shared grid), inverse over the 32x2 special limbs and over the 32x3 c1 limbs
    # -> 6 kernel launches (2 passes each)
    CH = 128  # the library's chunk size
    mix = [(CH * L * K + CH * L, False), (CH * 2, True), (CH * L, True)]
    npoly = sum(m[0] for m in mix)
    scratch = torch.zeros((CH * L * K + CH * L, n), dtype=torch.int64, device=dev)
    for cnt, inv in mix:
        X.ntt(scratch, cnt, 0, K, inv)
    torch.cuda.synchronize()
    reps = 50
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record(stream)
    for _ in range(reps):
        for cnt, inv in mix:
            X.ntt(scratch, cnt, 0, K, inv)
    k1.record(stream)
    torch.cuda.synchronize()
    launches = 2 * len(mix)
    ntt_ms = k0.elapsed_time(k1) / reps / launches           # average duration of one ntt_pass_kernel launch
    ntt_alg = 2 * npoly * n * 8 / launches                   # a transform reads and writes each polynomial once (2 passes)
    del scratch

    res = None
    if rank == 0:
        A = a_block_bytes(n, L, K)
        path_ms = dev_ms / args.steps
        traffic = None  # HBM-side bytes per launch of the path from rocprofv3 PMC passes (tools/pmc_traffic.py)
        tf = os.path.join(ROOT, "profiles", "r1_pmc_traffic_b256_final2.json")
        if os.path.exists(tf) and args.params == "config2":
            traffic = json.load(open(tf))["traffic_bytes_per_transciphering"] * B
        achieved = A * B / (path_ms * 1e-3) / 1e9
        res = {
            "metric": "PASTA-3 transcipherings/sec (N=2^15, 4 RNS limbs)" if args.params == "config2" else "PASTA-3 transcipherings/sec (N=2^14, BFVDefault 9 primes)", "value": value, "unit": "transcipherings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"reference defaults: N=2^14, BFVDefault 9 primes (L=8,K=9), t=65537, batch-{B} blocks per GPU") if args.params != "config2" else
                                   ("BASELINE config 2: N=2^15, coeff_modulus 4x60-bit (L=3,K=4), t=65537, "
                                    f"batch-{B} independent 128-word PASTA-3 blocks per GPU, block counter 0") if args.workload == "config2" else
                                   (f"MNIST-shaped: N=2^15, 4x60-bit, t=65537, batch-{B} blocks per GPU = 784-word samples x 7 block counters (last block 16 words)"),
                       "batch_per_gpu": B, "sharding": f"{world} rank(s), independent items, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": "profiles/r1_pmc_traffic_b256_final2.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                         "kernel": "hhe_pasta3_transcipher (whole path; SURVEY 8d A_block)",
                         "algorithmic_bytes_per_unit": A, "units_per_launch": B, "launch_ms": path_ms,
                         "dominant_kernel": {"name": "ntt_pass_kernel", "launch_mix": f"one rotation step at chunk {CH} through hhe_ntt: fwd {CH * 15} (digits + c0 limbs), inv {CH * 2}, inv {CH * 3} polys = 6 launches (the pipeline itself shares the inverse row pass: 5 launches)",
                                             "algorithmic_bytes_per_launch": ntt_alg, "avg_launch_us": ntt_ms * 1e3,
                                             "us_per_polynomial": ntt_ms * 1e3 * launches / npoly,
                                             "achieved_GBps": ntt_alg / (ntt_ms * 1e-3) / 1e9,
                                             "frac": ntt_alg / (ntt_ms * 1e-3) / 1e9 / 8000.0}},
        }
        if world == 1 and args.cpu_baseline:
            import oracle as orc  # CPU baseline leg only
            O = orc.Oracle(logn, q, t)
            threads = min(16, len(os.sched_getaffinity(0)))  # the GPU box grants 16 cores per GPU
            nb = threads * args.cpu_blocks_per_thread
            elts = sorted(gks)
            gk = orc.GaloisKeys(elts, np.stack([gks[e] for e in elts]))
            c0 = time.perf_counter()
            ref = O.transcipher_batch(enc_key, rk, gk, cw[:nb], ncw[:nb], bidx[:nb], threads=threads)
            cpu_s = time.perf_counter() - c0
            got = out[:1].cpu().numpy().view(np.uint64)
            res["cpu_baseline"] = {"value": nb / cpu_s, "unit": "transcipherings/s", "cores": threads, "kind": "port",
                                   "sample": f"{nb} blocks of the same workload ({args.cpu_blocks_per_thread}/thread, OpenMP over blocks), {cpu_s:.1f} s",
                                   "matches_gpu_item0": bool((got[0] == ref[0]).all())}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    X.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
