"""The C ABI of the batch split on the MI355X (include/nflhip.h "multi-GPU"): the shard-composable digest against its
numpy statement, peer copies and the one-process scatter / gather between contexts, and the RCCL communicator REALLY
initialised -- at world size 1, which is what a 1-GPU box can run: ncclCommInitRank, the degenerate scatter / gather (the
root's own shard), the one-word all-reduce barrier and the all-gather of digests."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.mark.parametrize("lb,n,nm,batch", [(64, 4096, 4, 19), (32, 1024, 2, 33), (16, 128, 1, 77), (64, 16, 3, 5), (16, 4, 1, 3)])
def test_digest_equals_the_numpy_statement_and_composes_over_shards(torch, lb, n, nm, batch):
    from nfllib_amd import Engine, sharding, shard_range
    eng = Engine(lb, n, nm)
    d = eng.fill_uniform(eng.empty(batch), 99, 0)
    h = eng.to_host(d)
    whole = eng.digest(d)
    assert whole == sharding.digest_words(h)
    per = n * nm
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            f, c = shard_range(batch, world, r)
            if c:
                got = eng.digest(d[f:f + c], first_poly=f)          # (slices of u16 rows of 4 words are 8-byte aligned only)
                assert got == sharding.digest_words(h[f:f + c], first_word=f * per)
                parts.append(got)
        assert sharding.combine_digests(parts) == whole
    eng.close()


def test_scatter_and_gather_between_contexts_of_one_process(torch):
    """two contexts (here: of the same device -- a 1-GPU box), each with its own stream: shards arrive, are worked on, and
    come home; and nflhip_memcpy_peer_dev between them"""
    from nfllib_amd import Engine, _lib, shard_range
    lib = _lib.lib
    e0, e1, e2 = Engine(64, 4096, 4), Engine(64, 4096, 4), Engine(64, 4096, 4)
    total = 11
    full = e0.fill_uniform(e0.empty(total), 5, 0)
    engs = [e0, e1, e2]
    streams = [torch.cuda.Stream() for _ in engs]
    shards = [e.empty(shard_range(total, 3, r)[1]) for r, e in enumerate(engs)]
    ctxs = (C.c_void_p * 3)(*[e.ctx for e in engs])
    ptrs = (C.c_void_p * 3)(*[s.data_ptr() for s in shards])
    sts = (C.c_void_p * 3)(*[s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    for root in (0, 2):
        assert lib.nflhip_scatter_local_dev(ctxs, 3, ptrs, root, C.c_void_p(full.data_ptr()), total, sts) == 0, lib.nflhip_last_error(None)
        torch.cuda.synchronize()
        for r in range(3):
            f, c = shard_range(total, 3, r)
            assert torch.equal(shards[r], full[f:f + c])
        home = torch.zeros_like(full)
        assert lib.nflhip_gather_local_dev(ctxs, 3, C.c_void_p(home.data_ptr()), root, ptrs, total, sts) == 0
        torch.cuda.synchronize()
        assert torch.equal(home, full)
    dst = torch.zeros_like(shards[1])
    assert lib.nflhip_memcpy_peer_dev(e1.ctx, C.c_void_p(dst.data_ptr()), e0.ctx, C.c_void_p(shards[1].data_ptr()), dst.numel() * 8, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst, shards[1])
    for e in engs:
        e.close()


def test_rccl_really_initialised_at_world_size_one(torch, monkeypatch):
    """ncclGetUniqueId + ncclCommInitRank(nranks = 1) through the C ABI, then everything the N-rank path calls"""
    from nfllib_amd import Comm, Engine
    eng = Engine(64, 4096, 4)
    uid = Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = Comm(eng, 1, 0, uid)
    assert comm.lib.nflhip_comm_rank(comm.h) == 0 and comm.lib.nflhip_comm_size(comm.h) == 1
    comm.barrier()
    assert comm.allgather_u64(0xDEADBEEFCAFEF00D) == [0xDEADBEEFCAFEF00D]
    total = 9
    full = eng.fill_uniform(eng.empty(total), 7, 0)
    shard = eng.empty(total)
    comm.scatter(shard, full, total)
    torch.cuda.synchronize()
    assert torch.equal(shard, full)
    c = eng.polymul(shard, shard)
    home = torch.zeros_like(full)
    comm.gather(home, c, total)
    torch.cuda.synchronize()
    assert torch.equal(home, eng.polymul(full, full))
    assert comm.allgather_u64(eng.digest(c)) == [eng.digest(home)]
    comm.close()
    eng.close()


def test_comm_argument_errors_are_status_codes(torch):
    from nfllib_amd import Comm, Engine, NflHipError
    eng = Engine(32, 1024, 1)
    uid = Comm.unique_id()
    with pytest.raises(NflHipError):
        Comm(eng, 2, 2, uid)                 # rank out of range
    comm = Comm(eng, 1, 0, uid)
    with pytest.raises(NflHipError):
        comm.scatter(eng.empty(1), eng.empty(1), 1, root=3)
    comm.close()
    eng.close()
