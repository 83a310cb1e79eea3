"""GPU suite, SURVEY §8 row a1: al3d_merge_sweeps_f32 through the C ABI against the reference golden
vector and the oracle (bit-exact), plus edge cases and a full-size frame."""
import numpy as np
import pytest
import torch

from test_sweeps_oracle import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_device_merge_equals_reference_golden():
    from al3d.datasets.nusc_files import merge_sweeps_device
    z, files, xf, tl = load_case()
    got = merge_sweeps_device(files, xf, tl, DEV).cpu().numpy()
    assert got.shape == z["combined"].shape
    assert np.array_equal(got.view(np.int32), z["combined"].view(np.int32))


def test_device_loader_from_files(tmp_path):
    from al3d.datasets import PoolFrames
    z, _, _, _ = load_case()
    n = len(z["time_lag"])
    for f in range(n + 1):
        z[f"raw{f}"].tofile(tmp_path / f"f{f}.bin")
    sweeps = [dict(lidar_path=f"f{1 + i}.bin", time_lag=float(z["time_lag"][i]),
                   transform_matrix=z["xform"][i] if z["has_xform"][i] else None) for i in range(n)]
    info = dict(lidar_path="f0.bin", sweeps=sweeps, token="tok0")
    pool = PoolFrames.from_files([info, info], DEV, nsweeps=3, root=str(tmp_path))
    assert len(pool) == 2 and pool.tokens == ["tok0", "tok0"]
    # list order (no rng): key + sweeps 0, 1
    from al3d.datasets.nusc_files import load_frame_points
    ref = load_frame_points(info, nsweeps=3, root=str(tmp_path))
    assert np.array_equal(pool.frames[0].cpu().numpy().view(np.int32), ref.view(np.int32))
    with pytest.raises(AssertionError):
        PoolFrames.from_files([info], DEV, nsweeps=10, root=str(tmp_path))


def test_device_merge_edges_and_full_size(oracle):
    from al3d.datasets.nusc_files import merge_sweeps_device
    empty = np.zeros((0, 5), dtype=np.float32)
    assert merge_sweeps_device([empty], [None], [0.0], DEV).shape == (0, 5)
    sw = np.array([[0.5, 0.5, 0, 7, 1], [0.5, 1.0, 0, 8, 2], [-1.0, 0.0, 3, 9, 3]], dtype=np.float32)
    T = np.eye(4)
    T[:3, 3] = [10, 20, 30]
    out = merge_sweeps_device([empty, sw], [None, T], [0.0, 0.25], DEV).cpu().numpy()
    assert out.tolist() == [[10.5, 21.0, 30.0, 8.0, 0.25], [9.0, 20.0, 33.0, 9.0, 0.25]]
    # nuScenes-sized frame: 10 files x ~34.7k points, random rigid transforms -> bit-exact vs oracle
    rng = np.random.default_rng(3)
    files, xf, tl = [], [], []
    for f in range(10):
        p = rng.normal(0, 20, size=(34720, 5)).astype(np.float32)
        p[:2000, :2] = rng.uniform(-1.5, 1.5, size=(2000, 2))
        files.append(p)
        a = rng.uniform(-3, 3)
        M = np.eye(4)
        M[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        M[:3, 3] = rng.uniform(-5, 5, 3)
        xf.append(None if f in (0, 4) else M)
        tl.append(0.05 * f)
    got = merge_sweeps_device(files, xf, tl, DEV).cpu().numpy()
    ref = oracle.merge_sweeps(files, xf, tl, 1.0)
    assert got.shape == ref.shape and np.array_equal(got.view(np.int32), ref.view(np.int32))
    assert np.all(got[:34720, 4] == 0) and got.shape[0] < 347200


def test_streaming_file_loader_equals_the_reference_loading_rules(tmp_path):
    """FileSweepLoader (native reader pool -> pinned staging -> batched device merge -> voxelizer) on a synthetic
    on-disk pool: every frame's merged cloud has the bits of the reference loader's rules
    (det3d/datasets/pipelines/loading.py:17-63,98-126 as restated in nusc_files.load_frame_points) and of the
    single-frame device path; ragged last batch, an empty sweep file, a file with a partial trailing row, a sweep
    without transform; the voxelized example equals DeviceSweepLoader's on the same clouds."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from write_synthetic_pool import write_pool
    from al3d.datasets import DeviceSweepLoader, FileSweepLoader, PoolFrames, generate_task_anchors
    from al3d.datasets.nusc_files import load_frame_points, load_frame_points_device
    from al3d.utils import Config
    root = str(tmp_path / "pool")
    infos, _ = write_pool(root, scenes=1, base=3, nsweeps=10)
    infos = infos[:7]                                                # 7 frames, batch 3 -> 3 + 3 + 1
    # edge cases inside real frames
    open(os.path.join(root, infos[1]["sweeps"][2]["lidar_path"]), "wb").close()          # an empty sweep file
    with open(os.path.join(root, infos[2]["sweeps"][0]["lidar_path"]), "ab") as f:       # partial trailing row
        f.write(b"\x00" * 13)
    infos[4]["sweeps"][5]["transform_matrix"] = None
    cfg = Config.fromfile(os.path.join(os.path.dirname(__file__), "..", "examples", "active", "cbgs_spatial_temporal.py"))
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    loader = FileSweepLoader(infos, cfg.voxel_generator, anchors, batch_size=3, device=DEV, root=root, threads=4, depth=2)
    assert len(loader) == 3
    clouds, examples = [], []
    for ex in loader:
        off = ex["point_offsets"].cpu().numpy()
        pts = ex["points"].cpu().numpy()
        assert off[0] == 0 and len(off) == len(ex["metadata"]) + 1
        for b in range(len(off) - 1):
            clouds.append(pts[off[b]:off[b + 1]])
        examples.append(ex)
    assert [m["index"] for ex in examples for m in ex["metadata"]] == list(range(7))
    for i, info in enumerate(infos):
        ref = load_frame_points(info, nsweeps=10, root=root)
        assert clouds[i].shape == ref.shape and np.array_equal(clouds[i].view(np.int32), ref.view(np.int32)), i
        one = load_frame_points_device(info, DEV, nsweeps=10, root=root).cpu().numpy()
        assert np.array_equal(one.view(np.int32), ref.view(np.int32))
    # same voxels as the resident-pool loader on the same clouds
    pool = PoolFrames.from_numpy(clouds, DEV)
    ref_ex = list(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=3, device=DEV))
    for a, b in zip(examples, ref_ex):
        assert torch.equal(a["coordinates"], b["coordinates"]) and torch.equal(a["voxel_features"], b["voxel_features"])
        assert torch.equal(a["num_points"], b["num_points"])
    # a second pass over the loader re-reads the files (buffers are recycled) and gives the same bits
    again = torch.cat([ex["points"] for ex in loader])
    assert torch.equal(again, torch.cat([ex["points"] for ex in examples]))
    assert loader.bytes_read > 0
