"""Work partitioning over one-process-per-GPU ranks.

The path shards by independent units ((sample, block) transcipherings; a sample's blocks stay on one
GPU so flatten needs no exchange -- SURVEY 8e).  There is no data-path collective: ranks only meet at a
barrier and a MAX-reduce of the elapsed time used for reporting.
"""
import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_samples(n_samples, rank, world):
    """contiguous sample range [lo, hi) of this rank (balanced to within one sample)."""
    base, rem = divmod(n_samples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_blocks(sample_words):
    """(block counter, words in block) pairs of one sample of `sample_words` symmetric-ciphertext words"""
    nb = (sample_words + 127) // 128
    return [(b, min(128, sample_words - 128 * b)) for b in range(nb)]


def work_items(lo, hi, sample_words):
    """flat (sample, block, nwords) list for samples [lo, hi)"""
    return [(s, b, w) for s in range(lo, hi) for (b, w) in sample_blocks(sample_words)]


def init_process_group(backend):
    import torch.distributed as dist
    rank, world, _ = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_max_sum(elapsed_s, units, device="cpu"):
    """whole-job view: (max elapsed over ranks, total units over ranks)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def checksum_words(words):
    """order-independent 64-bit checksum of a uint64 array (for cross-rank result accounting)"""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    return int(np.bitwise_xor.reduce(w)) ^ (int(w.sum(dtype=np.uint64)) << 1 & 0xFFFFFFFFFFFFFFFF)
