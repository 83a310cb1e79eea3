"""CPU suite, SURVEY §8 row a1: the oracle's sweep merge and the host numpy reader against the golden
vector produced by the reference's own LoadPointCloudFromFile (oracle/gen_golden_sweeps.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sweeps.npz")


def load_case():
    z = np.load(GOLD, allow_pickle=False)
    order = z["order"].tolist()
    nfiles = 1 + len(z["time_lag"])
    raws = []
    for f in range(nfiles):
        r = z[f"raw{f}"]
        raws.append(r[: r.shape[0] - r.shape[0] % 5].reshape(-1, 5))
    files = [raws[0]] + [raws[1 + i] for i in order]
    xf = [None] + [z["xform"][i] if z["has_xform"][i] else None for i in order]
    tl = [0.0] + [float(z["time_lag"][i]) for i in order]
    return z, files, xf, tl


def test_oracle_merge_equals_reference(oracle):
    z, files, xf, tl = load_case()
    got = oracle.merge_sweeps(files, xf, tl, 1.0)
    assert got.shape == z["combined"].shape
    assert np.array_equal(got.view(np.int32), z["combined"].view(np.int32))


def test_host_reader_equals_reference(tmp_path):
    """al3d.datasets.nusc_files.load_frame_points (numpy, the reference's own formulas) on the same files."""
    from al3d.datasets.nusc_files import load_frame_points
    z, _, _, _ = load_case()
    n = len(z["time_lag"])
    for f in range(n + 1):
        z[f"raw{f}"].tofile(tmp_path / f"f{f}.bin")
    sweeps = [dict(lidar_path=f"f{1 + i}.bin", time_lag=float(z["time_lag"][i]),
                   transform_matrix=z["xform"][i] if z["has_xform"][i] else None) for i in range(n)]

    class Replay:                                   # the order the reference drew (np.random.choice)
        def choice(self, n_, k, replace=False):
            return z["order"]
    pts = load_frame_points(dict(lidar_path="f0.bin", sweeps=sweeps), nsweeps=len(z["order"]) + 1,
                            root=str(tmp_path), rng=Replay())
    assert np.array_equal(pts.view(np.int32), z["combined"].view(np.int32))


def test_oracle_merge_edges(oracle):
    key = np.zeros((0, 5), dtype=np.float32)
    sw = np.array([[0.5, 0.5, 0, 7, 1], [0.5, 1.0, 0, 8, 2], [-1.0, 0.0, 3, 9, 3]], dtype=np.float32)
    T = np.eye(4)
    T[:3, 3] = [10, 20, 30]
    out = oracle.merge_sweeps([key, sw], [None, T], [0.0, 0.25], 1.0)
    # first row is inside the 1 m square and dropped; |y| == 1 and |x| == 1 are kept (strict <)
    assert out.tolist() == [[10.5, 21.0, 30.0, 8.0, 0.25], [9.0, 20.0, 33.0, 9.0, 0.25]]
    assert oracle.merge_sweeps([key], [None], [0.0]).shape == (0, 5)
