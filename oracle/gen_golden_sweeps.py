"""TEST INFRASTRUCTURE ONLY.  Golden vectors for SURVEY §8 row a1 (sweep merge).

Runs the reference's own ``LoadPointCloudFromFile`` (NuScenesDataset branch, with ``read_file`` /
``remove_close`` / ``read_sweep``; det3d/datasets/pipelines/loading.py:17-126) on synthetic ``.bin``
files written to a temp dir, and commits inputs + outputs as ``tests/golden/sweeps.npz``.
The reference draws the sweep order from ``np.random.choice``; the same draw is replayed here to
record the order it used.  Run in the build container only (needs /root/reference).
"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
import oracle  # noqa: E402


def import_loading():
    R.install_standins()
    bno = R.import_box_np_ops()
    import det3d.torchie  # noqa: F401
    sys.modules["det3d.core"].box_np_ops = bno
    for pkg in ("det3d.datasets", "det3d.datasets.pipelines", "det3d.datasets.kitti"):
        if pkg not in sys.modules:
            R._pkg(pkg, os.path.join(R.REFERENCE_ROOT, *pkg.split(".")))
    pm = R._mod("pycocotools")
    pm.mask = R._mod("pycocotools.mask")          # named at import, unused on this path
    kc = R._mod("det3d.datasets.kitti.kitti_common")   # needs skimage at import; unused on this path
    sys.modules["det3d.datasets.kitti"].kitti_common = kc
    return importlib.import_module("det3d.datasets.pipelines.loading")


def rigid(rng):
    yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-0.02, 0.02)
    cz, sz, cy, sy = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry
    T[:3, 3] = rng.uniform(-6, 6, 3) * [1, 1, 0.05]
    return T


def main():
    loading = import_loading()
    rng = np.random.default_rng(11)
    tmp = tempfile.mkdtemp(prefix="al3d_sweeps_")
    nfiles, nsweeps = 6, 5                       # key + 5 candidate sweeps, 4 of them used
    raws = []
    for f in range(nfiles):
        p = int(rng.integers(300, 700))
        pts = np.empty((p, 5), dtype=np.float32)
        pts[:, :2] = rng.normal(0, 12, (p, 2))
        pts[: p // 6, :2] = rng.uniform(-1.6, 1.6, (p // 6, 2))     # a cluster around the sensor
        pts[0, :2] = [1.0, 0.25]                                    # |x| == radius is kept (strict <)
        pts[1, :2] = [-0.999999, 0.999999]
        pts[:, 2] = rng.normal(-1, 0.8, p)
        pts[:, 3] = rng.integers(0, 256, p)
        pts[:, 4] = rng.integers(0, 32, p)
        raw = pts.reshape(-1)
        if f == 2:
            raw = np.concatenate([raw, np.float32([1.0, 2.0, 3.0])])   # trailing partial row is dropped
        path = os.path.join(tmp, f"f{f}.bin")
        raw.astype(np.float32).tofile(path)
        raws.append(pts)
    sweeps = []
    for f in range(1, nfiles):
        sweeps.append(dict(lidar_path=os.path.join(tmp, f"f{f}.bin"),
                           transform_matrix=None if f == 3 else rigid(rng),
                           time_lag=float(0.05 * f + rng.uniform(0, 0.01))))
    info = dict(lidar_path=os.path.join(tmp, "f0.bin"), sweeps=sweeps)
    np.random.seed(5)
    order = np.random.choice(len(sweeps), nsweeps - 1, replace=False)
    np.random.seed(5)
    res = {"lidar": {"nsweeps": nsweeps}}
    loading.LoadPointCloudFromFile(dataset="NuScenesDataset")(res, info)
    combined = res["lidar"]["combined"]
    assert combined.dtype == np.float32 and combined.shape[1] == 5
    # the restatement must reproduce the reference bit for bit on this fixture
    files = [raws[0]] + [raws[1 + i] for i in order]
    xf = [None] + [sweeps[i]["transform_matrix"] for i in order]
    tl = [0.0] + [sweeps[i]["time_lag"] for i in order]
    mine = oracle.merge_sweeps(files, xf, tl, 1.0)
    same = mine.shape == combined.shape and np.array_equal(mine.view(np.int32), combined.view(np.int32))
    print("rows", combined.shape[0], "order", order.tolist(), "oracle == reference bit-exact:", same)
    assert same, "float64 evaluation order differs from numpy's dot on this fixture"
    out = {"order": order.astype(np.int64), "combined": combined,
           "time_lag": np.array([s["time_lag"] for s in sweeps], dtype=np.float64),
           "has_xform": np.array([s["transform_matrix"] is not None for s in sweeps], dtype=np.uint8),
           "xform": np.stack([np.eye(4) if s["transform_matrix"] is None else s["transform_matrix"] for s in sweeps])}
    for f in range(nfiles):
        out[f"raw{f}"] = np.fromfile(os.path.join(tmp, f"f{f}.bin"), dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "sweeps.npz"), **out)


if __name__ == "__main__":
    main()
