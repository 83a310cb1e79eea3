"""GPU suite: the N>1 selection path with two ranks sharing cuda:0 (gloo transport, device
tensors): sharded all-pairs geodesics + all-gather must reproduce the single-process map bit for
bit, and the gathered embeddings must come back in dataset order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from al3d import selector_ops as ops, synthetic
        from al3d.sweep import gather_in_dataset_order
        dev = torch.device("cuda:0")
        infos, _ = synthetic.make_pool(110, seed=3)            # N = 4400 >= the sharding threshold
        cfgm, _, _ = synthetic.pool_arrays(infos)
        xy = torch.from_numpy(np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfgm])).to(dev)
        S = ops.spatial_map(xy, 8)                              # sharded over the two ranks
        d, i = ops.knn_2d(xy, 9)
        S1 = ops.apsp_knn(d, i)                                 # single-process reference
        ok_map = bool(torch.equal(S.view(torch.int64), S1.view(torch.int64)))
        # feature map: row blocks per rank + all-gather == the single-process map, bit for bit
        from al3d import lib
        g = torch.Generator(device="cpu").manual_seed(5)
        feats = torch.randn(xy.shape[0], 48, generator=g).to(dev)
        F = ops.l1_distance(feats, 2)
        F1 = torch.empty_like(F)
        lib.call("al3d_l1_distance_f32", feats.data_ptr(), feats.shape[0], feats.shape[1], 2, F1.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
        ok_map = ok_map and bool(torch.equal(F.view(torch.int32), F1.view(torch.int32)))
        n = 37
        full = torch.arange(n * 4, dtype=torch.float32, device=dev).view(n, 4)
        per = (n + world - 1) // world
        idx = torch.arange(rank * per, min(n, (rank + 1) * per), device=dev)
        ok_gather = bool(torch.equal(gather_in_dataset_order(full[idx], idx, n), full))
        q.put((rank, ok_map, ok_gather))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == [(0, True, True), (1, True, True)]


def _select_worker(rank, world, port, tmp, q):
    """Both ranks run the whole SpatialTemporalFeatureSelector selection (as bench.py does under
    torch.distributed): geodesic map prefetched on a side stream with its rows sharded + gathered,
    feature map row-sharded + gathered, greedy replicated."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from al3d.selectors import build_selector
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = dict(type="SpatialTemporalFeatureSelector", budget=120, buffer_file=os.path.join(tmp, "buffer.json"),
                   infos_origin=os.path.join(tmp, "infos.pkl"), logs_file=os.path.join(tmp, "log.json"),
                   buffer_path=os.path.join(tmp, "feats.pt"), distance_store_file=None, pred=False)
        random.seed(3407)
        sel = build_selector(cfg)
        sel.prefetch_spatial_map(dev)                 # what _SweepMixin.buffer_pred does before the sweep
        sel.select_samples(local_rank=0)
        q.put((rank, sel.selected_index[sel.current_budget]))
    finally:
        dist.destroy_process_group()


def test_three_rank_selection_equals_oracle(oracle, tmp_path):
    import json
    import pickle
    import random
    from al3d import synthetic
    tmp = str(tmp_path)
    infos, logs = synthetic.make_pool(110, seed=4)                  # N = 4400 -> sharded maps
    n = len(infos)
    feats = synthetic.make_embeddings(n, seed=2, scale=0.01)[:, :64].copy()
    pickle.dump(infos, open(os.path.join(tmp, "infos.pkl"), "wb"))
    json.dump(logs, open(os.path.join(tmp, "log.json"), "w"))
    json.dump({"0": []}, open(os.path.join(tmp, "buffer.json"), "w"))
    torch.save(torch.from_numpy(feats), os.path.join(tmp, "feats.pt"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3                                                       # 4400 rows over 3 ranks: uneven blocks
    procs = [ctx.Process(target=_select_worker, args=(r, world, port, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfgm, run_id, n_boxes = synthetic.pool_arrays(infos)
    xy = np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfgm])
    random.seed(3407)
    first = random.choice(range(n))
    D = oracle.combine(n, spatial=oracle.spatial_map(xy, 8), temporal_id=run_id, feat=oracle.l1_map_f32(feats, 2),
                       normalize="exp", aggregate="sum", lambda_t=1.0, lambda_f=1.0)
    rc, picks = oracle.greedy(D, [], first, np.array([int(b) * 0.04 for b in n_boxes]), 0.12, 0.0, 120.0)
    assert rc == 0
    assert all(results[r] == picks.tolist() for r in range(world))


def _rccl_worker(port, q):
    """One rank, backend nccl (= RCCL on ROCm): the collectives of the N > 1 sweep -- the row-count all_gather, the two
    all_gather_into_tensor of embeddings and indices, the ranks_seen all_reduce of bench.py -- on device tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from al3d.sweep import gather_in_dataset_order
        dev = torch.device("cuda:0")
        n = 1003
        full = torch.randn(n, 512, generator=torch.Generator().manual_seed(1)).to(dev)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(2)).to(dev)
        out = gather_in_dataset_order(full[idx], idx, n, collective_at_world_1=True)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        dist.barrier()
        torch.cuda.synchronize()
        q.put((dist.get_backend(), bool(torch.equal(out, full)), float(ones.item())))
    finally:
        dist.destroy_process_group()


def test_rccl_carries_the_sweeps_collectives_on_one_gpu():
    """The N > 1 data path has only ever been rehearsed over gloo (no multi-GPU box for the builder).  This runs its exact
    collective calls through RCCL itself in a one-rank group on the box's GPU: communicator set-up, all_gather,
    all_gather_into_tensor, all_reduce and barrier on device memory; the gathered tensor must be the dataset-ordered one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    backend, ok, seen = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert backend == "nccl" and ok and seen == 1.0
