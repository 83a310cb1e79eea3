"""GPU suite: the reference's val pipeline (LoadPointCloudFromFile -> LoadPointCloudAnnotations ->
Preprocess -> Voxelization -> AssignTarget -> Reformat, examples/active/cbgs_*.py:295-302) built
from the PIPELINES registry with device-backed stages, collated and fed to the detector."""
import hashlib
import os

import numpy as np
import pytest
import torch

from test_sweeps_oracle import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _infos(tmp_path, z, copies=2):
    n = len(z["time_lag"])
    for f in range(n + 1):
        z[f"raw{f}"].tofile(tmp_path / f"f{f}.bin")
    sweeps = [dict(lidar_path=str(tmp_path / f"f{1 + i}.bin"), time_lag=float(z["time_lag"][i]),
                   transform_matrix=z["xform"][i] if z["has_xform"][i] else None) for i in range(n)]
    return [dict(lidar_path=str(tmp_path / "f0.bin"), sweeps=sweeps, token=f"tok{k}") for k in range(copies)]


def test_val_pipeline_matches_reference_stage_by_stage(oracle, tmp_path):
    from al3d.datasets import PIPELINES, SweepDataset, collate_device
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "examples", "active", "cbgs_spatial_temporal.py"))
    assert {t["type"] for t in cfg.test_pipeline} <= set(PIPELINES.module_dict)
    z, _, _, _ = load_case()
    infos = _infos(tmp_path, z)
    ds = SweepDataset(infos, cfg.test_pipeline, nsweeps=len(z["order"]) + 1, class_names=cfg.class_names)
    np.random.seed(5)                                   # the draw the reference made for the golden
    ex0 = ds[0]
    # a1: the loaded cloud equals the reference's res["lidar"]["combined"]
    assert np.array_equal(ex0["points"].cpu().numpy().view(np.int32), z["combined"].view(np.int32))
    # a2: voxels of that cloud equal the oracle voxelizer (itself pinned to points_to_voxel_new)
    vg = cfg.voxel_generator
    rng = np.asarray(vg["range"], dtype=np.float32)
    vs = np.asarray(vg["voxel_size"], dtype=np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    v, c, n, f = oracle.voxelize(z["combined"], rng[:3], vs, grid, vg["max_points_in_voxel"], vg["max_voxel_num"])
    assert np.array_equal(ex0["coordinates"].cpu().numpy(), c)
    assert np.array_equal(ex0["num_points"].cpu().numpy(), n.astype(np.int64))
    assert np.array_equal(ex0["voxels"].cpu().numpy().view(np.int32), v.view(np.int32))
    assert ex0["num_voxels"].tolist() == [c.shape[0]] and list(ex0["shape"]) == [1024, 1024, 40]
    # a3: anchors equal the reference golden (digest of the full arrays)
    g = np.load(os.path.join(ROOT, "tests", "golden", "anchors.npz"), allow_pickle=False)
    assert len(ex0["anchors"]) == 6
    for t, a in enumerate(ex0["anchors"]):
        a = a.cpu().numpy()
        assert a.shape[0] == int(g[f"task{t}_count"])
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g[f"task{t}_sha256"])
    # a4 + detector: collate two samples and run the sweep forward
    np.random.seed(5)
    batch = collate_device([ex0, ds[1]])
    assert batch["coordinates"].shape[1] == 4 and batch["coordinates"].dtype == torch.int32
    assert int(batch["coordinates"][:, 0].max()) == 1 and batch["num_voxels"].tolist() == [c.shape[0]] * 2
    from al3d import synthetic
    from al3d.models import build_detector
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    with torch.no_grad():
        preds, middle = model(batch, return_loss=False, estimate=True)
        emb = middle[-1].mean(-1).mean(-1)
        # the reader path (voxels + num_points -> VoxelFeatureExtractorV3) gives the same embedding
        batch2 = {k: v for k, v in batch.items() if k != "voxel_features"}
        _, middle2 = model(batch2, return_loss=False, estimate=True)
        emb2 = middle2[-1].mean(-1).mean(-1)
    assert emb.shape == (2, 512) and torch.isfinite(emb).all() and len(preds) == 2
    assert torch.equal(emb[0], emb[1])                   # two copies of the same frame
    assert torch.allclose(emb, emb2, rtol=1e-5, atol=1e-6)
