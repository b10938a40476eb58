"""Batch split of independent polynomials across the GPUs of one node.

The reference has no distributed layer: the only
multi-device strategy of this engine is the contiguous batch split of SURVEY.md
8(e) -- rank g of G owns polys [g*B/G, (g+1)*B/G) of the dense
[batch][NbModuli][Degree] tensor, every rank builds identical device tables
locally, and there is NO data-path collective.  torch.distributed (RCCL on the
GPUs, gloo in the CPU tests) is used only for the barrier, the max-over-ranks
clock and to combine per-shard digests ("checksum of checksums").
"""
import os

import numpy as np

MASK64 = (1 << 64) - 1


def shard_range(global_batch, world_size, rank):
    """Contiguous, balanced split: first (global_batch % world) ranks get one extra poly."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def digest_words(words, first_word=0):
    """Order-sensitive 64-bit digest of a word array that composes over shards:
    sum_i (i+1+first_word)*odd_mix(w_i) mod 2^64, so the digest of a
    concatenation is the sum of the digests of the parts."""
    w = np.ascontiguousarray(words).reshape(-1).astype(np.uint64)
    idx = (np.arange(w.size, dtype=np.uint64) + np.uint64(first_word + 1))
    with np.errstate(over="ignore"):
        z = (w ^ (w >> np.uint64(31))) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03)
        return int((idx * z).sum(dtype=np.uint64))


def combine_digests(parts):
    s = 0
    for p in parts:
        s = (s + int(p)) & MASK64
    return s


def allreduce_max(value, dist, device=None):
    """max over ranks of a python float (the timing contract of bench.py)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_digests(value, dist, world_size, device=None):
    import torch
    v = int(value)
    mine = torch.tensor([v & 0xFFFFFFFF, v >> 32], dtype=torch.int64, device=device if device is not None else "cpu")
    outs = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(outs, mine)
    return [int(o[0].item()) | (int(o[1].item()) << 32) for o in outs]
