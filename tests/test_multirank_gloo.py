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


def test_single_process_gather_is_a_scatter():
    from al3d.sweep import gather_in_dataset_order
    f = torch.rand(5, 3)
    idx = torch.tensor([3, 0, 4, 1, 2])
    out = gather_in_dataset_order(f, idx, 5)
    assert torch.equal(out[idx], f)
