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
