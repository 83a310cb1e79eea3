"""CPU suite: oracle voxelizer (+mean VFE) against the reference's points_to_voxel_new
golden vectors (oracle/gen_golden_detector.py)."""
import hashlib
import os

import numpy as np
import pytest

from al3d import synthetic

G = os.path.join(os.path.dirname(__file__), "golden")
RANGE_MIN = [-51.2, -51.2, -5.0]
VSIZE = [0.1, 0.1, 0.2]
GRID = [1024, 1024, 40]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    z = np.load(os.path.join(G, f"voxel_{name}.npz"))
    fx = {k: z[k] for k in z.files}
    pts = synthetic.make_point_cloud(int(fx["frame"]), nsweeps=int(fx["nsweeps"]))
    assert sha(pts) == str(fx["points_sha256"]), "synthetic.make_point_cloud drifted"
    return fx, pts


def check_against_golden(fx, voxels, coords_zyx, num, feat):
    assert len(coords_zyx) == int(fx["nvox"])
    assert sha(coords_zyx.astype(np.int32)) == str(fx["coords_sha256"])     # integer: exact
    assert sha(num.astype(np.int32)) == str(fx["num_sha256"])
    if voxels is not None:
        assert sha(voxels) == str(fx["voxels_sha256"])                       # copies: exact
    # mean VFE is floating point: torch's reduction order over the 10 slots is not the
    # slot order for large tensors, so the bound is a few float32 ulps, stated here.
    if "feat" in fx:
        np.testing.assert_allclose(feat, fx["feat"], rtol=2e-6, atol=1e-6)
    else:
        sl = slice(0, None, max(1, len(coords_zyx) // 512))
        assert np.array_equal(coords_zyx[sl], fx["coords_slice"])
        np.testing.assert_allclose(feat[sl], fx["feat_slice"], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["small", "capped", "full"])
def test_oracle_voxelizer_matches_reference(oracle, name):
    fx, pts = load(name)
    voxels, coords, num, feat = oracle.voxelize(pts, RANGE_MIN, VSIZE, GRID, int(fx["max_points"]),
                                                int(fx["max_voxels"]))
    check_against_golden(fx, voxels, coords, num, feat)


def test_oracle_voxelizer_edge_cases(oracle):
    # empty cloud, all points out of range, NaN coordinates
    e = np.zeros((0, 5), dtype=np.float32)
    v, c, n, f = oracle.voxelize(e, RANGE_MIN, VSIZE, GRID, 10, 100)
    assert len(c) == 0
    far = np.full((7, 5), 1e4, dtype=np.float32)
    far[3, 0] = np.nan
    v, c, n, f = oracle.voxelize(far, RANGE_MIN, VSIZE, GRID, 10, 100)
    assert len(c) == 0
    # the upper range edge is exclusive, the lower inclusive
    p = np.array([[-51.2, -51.2, -5.0, 1, 0], [51.2, 0, 0, 1, 0], [51.19999, 51.19999, 2.9999, 2, 0]],
                 dtype=np.float32)
    v, c, n, f = oracle.voxelize(p, RANGE_MIN, VSIZE, GRID, 10, 100)
    assert c.tolist() == [[0, 0, 0], [39, 1023, 1023]]
