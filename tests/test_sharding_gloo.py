"""CPU, world_size 2, gloo: the batch split of SURVEY.md 8(e).  Each rank runs the
hot path (here: the CPU oracle standing in for its GPU) on ITS shard of one logical
batch generated from the shared counter stream; the combined checksum-of-checksums
must equal the single-process result, and the bench timing reduction is a MAX."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4E464C6C6962
SHAPE = (64, 256, 3)
GLOBAL_BATCH = 7  # ragged on purpose: 4 + 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nfllib_amd import sharding
    from nfllib_amd.params import params
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, w, lr = sharding.env_rank_world()
    assert (r, w, lr) == (rank, world, rank)
    lb, n, m = SHAPE
    o = O.Oracle(lb, n, m, params(lb))
    lo, hi = sharding.shard_range(GLOBAL_BATCH, world, rank)
    a = o.fill_uniform(hi - lo, SEED, 0, first_poly=lo)
    b = o.fill_uniform(hi - lo, SEED, 1, first_poly=lo)
    c = o.polymul(a, b)
    dig = sharding.digest_words(c, first_word=lo * n * m)
    parts = sharding.allgather_digests(dig, dist, world)
    tmax = sharding.allreduce_max(1.0 + rank, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, sharding.combine_digests(parts), tmax))


def test_two_rank_batch_split_matches_single_process():
    sys.path.insert(0, ROOT)
    from nfllib_amd import sharding
    from nfllib_amd.params import params
    from oracle import oracle as O
    lb, n, m = SHAPE
    o = O.Oracle(lb, n, m, params(lb))
    a, b = o.fill_uniform(GLOBAL_BATCH, SEED, 0), o.fill_uniform(GLOBAL_BATCH, SEED, 1)
    want = sharding.digest_words(o.polymul(a, b))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    assert all(r[3] == want for r in res), "checksum of checksums differs from the single-process digest"
    assert all(r[4] == 2.0 for r in res), "timing reduction must be the max over ranks"


def _worker_sg(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from nfllib_amd import sharding
    from nfllib_amd.params import params
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lb, n, m = SHAPE
    o = O.Oracle(lb, n, m, params(lb))
    lo, hi = sharding.shard_range(GLOBAL_BATCH, world, rank)
    fa = fb = fc = None
    if rank == 0:  # the whole batch originates on the root only
        fa = torch.from_numpy(o.fill_uniform(GLOBAL_BATCH, SEED, 0).view(np.int64))
        fb = torch.from_numpy(o.fill_uniform(GLOBAL_BATCH, SEED, 1).view(np.int64))
        fc = torch.zeros_like(fa)
    # one operand in one message per peer, the other cut into many pieces (the cap that keeps a 16 GiB config-D shard
    # away from 32-bit counts, here lowered so that a few polynomials already need several groups; 5000 is not a
    # multiple of anything in sight)
    sa = sharding.scatter_batch(fa, torch.empty((hi - lo, m, n), dtype=torch.int64), dist, rank, world)
    sb = sharding.scatter_batch(fb, torch.empty((hi - lo, m, n), dtype=torch.int64), dist, rank, world, max_message_bytes=5000)
    c = o.polymul(sa.numpy().view(np.uint64), sb.numpy().view(np.uint64))
    sharding.gather_batch(torch.from_numpy(c.view(np.int64)), fc, dist, rank, world, max_message_bytes=7777)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sharding.digest_words(fc.numpy().view(np.uint64)) if rank == 0 else None))


def test_scatter_from_root_and_gather_back():
    """SURVEY.md 8(e) "when data originates on one device": root scatters contiguous shards with grouped
    point-to-point sends, every rank multiplies its shard, root gathers the products."""
    sys.path.insert(0, ROOT)
    from nfllib_amd import sharding
    from nfllib_amd.params import params
    from oracle import oracle as O
    lb, n, m = SHAPE
    o = O.Oracle(lb, n, m, params(lb))
    want = sharding.digest_words(o.polymul(o.fill_uniform(GLOBAL_BATCH, SEED, 0), o.fill_uniform(GLOBAL_BATCH, SEED, 1)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sg, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == want


def test_shard_range_properties():
    from nfllib_amd.sharding import shard_range
    for B in (0, 1, 7, 8, 1 << 20):
        for W in (1, 2, 3, 4, 8):
            spans = [shard_range(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_digest_composes_over_shards():
    from nfllib_amd.sharding import combine_digests, digest_words
    w = np.random.default_rng(0).integers(0, 2**63, size=1000, dtype=np.uint64)
    whole = digest_words(w)
    parts = [digest_words(w[:300]), digest_words(w[300:], first_word=300)]
    assert combine_digests(parts) == whole
    w2 = w.copy(); w2[[1, 2]] = w2[[2, 1]]
    assert digest_words(w2) != whole  # order-sensitive
