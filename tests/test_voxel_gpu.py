"""GPU suite: HIP voxelizer + mean VFE against the CPU oracle (bit-exact: integer
coordinates/counts, copied point slots, and the same slot-order float32 mean) and the
reference's golden vectors."""
import numpy as np
import pytest
import torch

from al3d import synthetic
from test_voxel_oracle import GRID, RANGE_MIN, VSIZE, check_against_golden, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def run_hip(frames, max_points, max_voxels, want_voxels=True):
    from al3d import detector_ops as D
    vox = D.Voxelizer(RANGE, VSIZE, max_points, max_voxels, max_batch=max(1, len(frames)), device=DEV)
    off = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).astype(np.int64)
    pts = np.concatenate(frames, axis=0) if len(frames) else np.zeros((0, 5), np.float32)
    outs = []
    for _ in range(2):          # second call checks that the persistent grid was left clean
        outs.append(vox(torch.from_numpy(pts).to(DEV), torch.from_numpy(off).to(DEV),
                        want_voxels=want_voxels))
    a, b = outs
    for k in ("feat", "coords", "num_points"):
        assert torch.equal(a[k], b[k]), k
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in a.items()}


@pytest.mark.parametrize("name", ["small", "capped", "full"])
def test_hip_voxelizer_matches_reference_and_oracle(oracle, name):
    fx, pts = load(name)
    mp, mv = int(fx["max_points"]), int(fx["max_voxels"])
    out = run_hip([pts], mp, mv)
    assert np.all(out["coords"][:, 0] == 0)
    check_against_golden(fx, out["voxels"], out["coords"][:, 1:], out["num_points"], out["feat"])
    v, c, n, f = oracle.voxelize(pts, RANGE_MIN, VSIZE, GRID, mp, mv)
    assert np.array_equal(out["coords"][:, 1:], c) and np.array_equal(out["num_points"], n)
    assert np.array_equal(out["voxels"].view(np.int32), v.view(np.int32))
    assert np.array_equal(out["feat"].view(np.int32), f.view(np.int32))


def test_hip_voxelizer_batch_concatenation(oracle):
    """Three ragged frames in one launch == three single-frame oracle runs concatenated with
    a batch index (collate_kitti semantics)."""
    frames = [synthetic.make_point_cloud(20, nsweeps=1), np.zeros((0, 5), np.float32),
              synthetic.make_point_cloud(21, nsweeps=2)]
    out = run_hip(frames, 10, 9000)
    row = 0
    for b, pts in enumerate(frames):
        v, c, n, f = oracle.voxelize(pts, RANGE_MIN, VSIZE, GRID, 10, 9000)
        m = len(c)
        assert out["num_voxels"][b] == m
        sl = slice(row, row + m)
        assert np.all(out["coords"][sl, 0] == b)
        assert np.array_equal(out["coords"][sl, 1:], c)
        assert np.array_equal(out["num_points"][sl], n)
        assert np.array_equal(out["feat"][sl].view(np.int32), f.view(np.int32))
        row += m
    assert row == len(out["coords"])


def test_hip_voxelizer_out_of_range_and_nan():
    far = np.full((7, 5), 1e4, dtype=np.float32)
    far[3, 0] = np.nan
    out = run_hip([far], 10, 100)
    assert len(out["coords"]) == 0
    p = np.array([[-51.2, -51.2, -5.0, 1, 0], [51.2, 0, 0, 1, 0], [51.19999, 51.19999, 2.9999, 2, 0]],
                 dtype=np.float32)
    out = run_hip([p], 10, 100)
    assert out["coords"].tolist() == [[0, 0, 0, 0], [0, 39, 1023, 1023]]


def test_hip_voxelizer_runs_of_identical_cells(oracle):
    """The wave-level aggregation of the voxelizer (one first-index atomic / grid probe / slot request per run of adjacent
    lanes in a cell) on the patterns that stress it: one cell for hundreds of consecutive points (runs across whole waves,
    far beyond max_points), two cells alternating point by point (no run at all, duplicates one lane apart from the
    run test), short runs, out-of-range points inside runs, and frame boundaries in the middle of a wave.  Bit-exact
    against the oracle per frame."""
    rng = np.random.default_rng(3)

    def cell_point(ix, iy, iz, k):
        base = np.array([-51.2 + 0.1 * ix, -51.2 + 0.1 * iy, -5.0 + 0.2 * iz], np.float32)
        jit = rng.uniform(0.01, 0.09, (k, 3)).astype(np.float32) * np.array([1, 1, 2], np.float32)
        return np.concatenate([base + jit, rng.uniform(0, 255, (k, 1)).astype(np.float32), np.zeros((k, 1), np.float32)], 1)

    a = cell_point(500, 500, 10, 300)                                        # one cell, 300 points in a row
    b = np.stack([cell_point(100, 200, 5, 1)[0] if i % 2 else cell_point(101, 200, 5, 1)[0] for i in range(150)])
    c = np.concatenate([cell_point(10 + i, 20, 3, int(r)) for i, r in enumerate(rng.integers(1, 7, 60))])
    d = cell_point(700, 300, 20, 40)
    d[5:9, 0] = 1e4                                                          # out of range inside a run
    d[20, 1] = np.nan
    f0 = np.concatenate([a, b, c, d]).astype(np.float32)                     # 300 + 150 + ~200 + 40 points: ends mid-wave
    f1 = np.concatenate([cell_point(500, 500, 10, 70), c[::-1], a[:33]]).astype(np.float32)   # same cells, next frame
    f2 = cell_point(1, 1, 1, 5)
    frames = [f0, f1, np.zeros((0, 5), np.float32), f2]
    out = run_hip(frames, 10, 9000)
    row = 0
    for bidx, pts in enumerate(frames):
        v, co, n, f = oracle.voxelize(pts, RANGE_MIN, VSIZE, GRID, 10, 9000)
        m = len(co)
        assert out["num_voxels"][bidx] == m
        sl = slice(row, row + m)
        assert np.all(out["coords"][sl, 0] == bidx)
        assert np.array_equal(out["coords"][sl, 1:], co)
        assert np.array_equal(out["num_points"][sl], n)
        assert np.array_equal(out["voxels"][sl].view(np.int32), v.view(np.int32))
        assert np.array_equal(out["feat"][sl].view(np.int32), f.view(np.int32))
        row += m
    assert row == len(out["coords"]) and out["num_points"].max() == 10


def test_points_tensor_longer_than_the_offsets():
    """A loader's compaction may leave unused rows behind the last frame (``points`` longer than ``point_offsets[-1]``): the
    rows past the end are not part of any frame -- same voxels as the truncated tensor, and the first-index grid is left
    clean (round 4: two passes used to walk the whole tensor and read per-point arrays nobody had written)."""
    import torch
    from al3d.detector_ops import Voxelizer
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-50, 50, (30000, 2)), rng.uniform(-4, 2, (30000, 1)), rng.uniform(0, 1, (30000, 2))], 1).astype(np.float32)
    tail = rng.uniform(-50, 50, (7000, 5)).astype(np.float32)
    vox = Voxelizer([-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.1, 0.1, 0.2], 10, 60000, max_batch=2, device="cuda:0")
    off = torch.tensor([0, 12000, 30000], dtype=torch.int64, device="cuda:0")
    a = vox(torch.from_numpy(pts).cuda(), off)
    for _ in range(3):                                                   # repeated calls: the grid must come back clean
        junk = torch.randint(-2**31, 2**31 - 1, (4_000_000,), dtype=torch.int64, device="cuda:0")   # dirty the allocator's blocks
        del junk
        b = vox(torch.from_numpy(np.concatenate([pts, tail])).cuda(), off)
        torch.cuda.synchronize()
        for k in ("feat", "coords", "num_points", "num_voxels"):
            assert torch.equal(a[k], b[k]), k
