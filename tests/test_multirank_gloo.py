"""CPU suite: the N>1 path of the sweep (frame shards -> all-gather -> dataset order) with two
gloo ranks.  Embeddings are plain tensors here; the collective and the order restoration are
the code under test (al3d.sweep.gather_in_dataset_order)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from al3d.sweep import gather_in_dataset_order
        full = torch.arange(n * 6, dtype=torch.float32).view(n, 6) * 0.5 + 1.0
        if mode == "contiguous":        # uneven contiguous blocks (what tools/active_select.py uses)
            per = (n + world - 1) // world
            idx = list(range(rank * per, min(n, (rank + 1) * per)))
        else:                           # the reference's DistributedSampler: rank::world, wrap-padded
            total = ((n + world - 1) // world) * world
            idx = (list(range(n)) + list(range(total - n)))[rank:total:world]
        idx_t = torch.tensor(idx, dtype=torch.int64)
        out = gather_in_dataset_order(full[idx_t], idx_t, n)
        q.put((rank, bool(torch.equal(out, full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,n", [("contiguous", 11), ("contiguous", 4), ("strided", 11), ("strided", 8)])
def test_two_rank_gather_restores_dataset_order(mode, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _worker8(rank, world, port, mode, n, empty_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from al3d.sweep import gather_in_dataset_order
        # row i holds a function of i only, so every rank can check the whole gathered tensor without a [n, 512] pool
        def rows(idx):
            i = idx.to(torch.float32)
            return torch.stack([i, i * 0.25 + 1.0, (idx % 97).to(torch.float32)], 1)
        if mode == "contiguous":
            per = (n + world - 1) // world
            idx = list(range(rank * per, min(n, (rank + 1) * per)))
        elif mode == "strided":
            total = ((n + world - 1) // world) * world
            idx = (list(range(n)) + list(range(total - n)))[rank:total:world]
        else:                               # one rank owns no frame at all (its shard of a short pool is empty)
            owners = [r for r in range(world) if r != empty_rank]
            idx = [] if rank == empty_rank else list(range(n))[owners.index(rank)::len(owners)]
        idx_t = torch.tensor(idx, dtype=torch.int64)
        out = gather_in_dataset_order(rows(idx_t), idx_t, n)
        q.put((rank, bool(torch.equal(out, rows(torch.arange(n))))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,n,empty_rank", [("contiguous", 28130, None), ("strided", 28130, None), ("with_empty", 1003, 5)])
def test_eight_rank_gather_at_the_full_pool_size(mode, n, empty_rank):
    """World 8 (BASELINE configs[2]: the full pool over 8 ranks) on CPU / gloo: 28,130 frames -- the real train split's
    size, not a multiple of 8 -- as contiguous shards (3,517 x 7 + 3,511) and as the reference's strided wrap-padded
    DistributedSampler shards, and a pool in which one rank owns no frame (zero-row tensors through both all-gathers)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, mode, n, empty_rank, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(8)]


def test_single_process_gather_is_a_scatter():
    from al3d.sweep import gather_in_dataset_order
    f = torch.rand(5, 3)
    idx = torch.tensor([3, 0, 4, 1, 2])
    out = gather_in_dataset_order(f, idx, 5)
    assert torch.equal(out[idx], f)
