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


# ---- moving a batch that lives on ONE rank (SURVEY.md 8(e) "Collective") ----------------------
# Steady state needs none of this (operands are generated in place on every device).  When a batch
# originates on one device, the root sends every peer ITS contiguous shard and later receives the
# result shards back: grouped point-to-point transfers (torch batch_isend_irecv = grouped
# ncclSend/ncclRecv on RCCL), one message per peer, so on xGMI every peer's shard travels over that
# peer's own direct link and nothing is reduced.  Tensors are moved as raw bytes (any limb width).
def _as_bytes(t):
    import torch
    return t.contiguous().view(torch.uint8).reshape(-1)


# One message per peer and group, at most this many bytes each: a config-D shard is 16 GiB, and 32-bit element counts
# inside a transport must never see it whole.  Pieces of one shard are posted in the same order on both sides, group
# after group (piece k of every peer travels in group k, so all links stay busy).
MAX_MESSAGE_BYTES = 1 << 30


def _pieces(buf, limit):
    n = buf.numel()
    return [buf[o:min(o + limit, n)] for o in range(0, n, limit)] if n else []


def _run_groups(dist, per_peer):
    """per_peer: list of (op, peer, [pieces]); issues group k = piece k of every peer that still has one."""
    depth = max((len(p) for _, _, p in per_peer), default=0)
    for k in range(depth):
        ops = [dist.P2POp(op, pieces[k], peer) for op, peer, pieces in per_peer if k < len(pieces)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def scatter_batch(full, shard, dist, rank, world, root=0, max_message_bytes=None):
    """Root holds `full` ([global_batch, nm, n]); every rank (root included) ends with its
    shard_range() slice of it in `shard` ([hi - lo, nm, n], preallocated, contiguous).  Returns `shard`."""
    if not shard.is_contiguous():
        raise ValueError("shard buffer must be contiguous")
    limit = max_message_bytes or MAX_MESSAGE_BYTES
    per_peer = []
    if rank == root:
        gb = full.shape[0]
        for r in range(world):
            lo, hi = shard_range(gb, world, r)
            if r == root:
                shard.copy_(full[lo:hi])
            elif hi > lo:
                per_peer.append((dist.isend, r, _pieces(_as_bytes(full[lo:hi]), limit)))
    elif shard.numel():
        per_peer.append((dist.irecv, root, _pieces(_as_bytes(shard), limit)))
    _run_groups(dist, per_peer)
    return shard


def gather_batch(shard, full, dist, rank, world, root=0, max_message_bytes=None):
    """Inverse of scatter_batch: the root's `full` (contiguous) receives every rank's shard at its slice."""
    limit = max_message_bytes or MAX_MESSAGE_BYTES
    per_peer = []
    if rank == root:
        if not full.is_contiguous():
            raise ValueError("destination batch must be contiguous")
        gb = full.shape[0]
        for r in range(world):
            lo, hi = shard_range(gb, world, r)
            if r == root:
                full[lo:hi].copy_(shard)
            elif hi > lo:
                buf = _as_bytes(full[lo:hi])   # a contiguous slice: the view aliases `full`
                assert buf.data_ptr() == full[lo:hi].data_ptr()
                per_peer.append((dist.irecv, r, _pieces(buf, limit)))
    elif shard.numel():
        per_peer.append((dist.isend, root, _pieces(_as_bytes(shard), limit)))
    _run_groups(dist, per_peer)
    return full if rank == root else None
