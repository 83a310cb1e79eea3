"""GPU suite: BASELINE configs[4] as a SWEEP -- the registered ``BEVFusion`` detector (Swin-T camera branch + voxelnet_0p075
lidar branch + ConvFuser + SECOND / SECONDFPN decoder [+ TransFusionHead]) behind the det3d detector contract, fed by
``CameraLidarSweepLoader``, swept by ``sweep_embeddings`` and driven by the selectors exactly like the CBGS detector
(tools/active_select.py:152-163 call sequence).  Seeded weights, synthetic six-camera frames and lidar clouds: parity unpinned
(mmcv / mmdet absent, no checkpoint); what is checked: contract, determinism, batch-size invariance of the fused-BEV embeddings,
pipeline modes, that both modalities reach the embedding, and the selectors' bookkeeping on top."""
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(name):
    from al3d import synthetic
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "examples", "active", name))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model.lidar, seed=0)
    for i, m in enumerate((model.camera_backbone, model.camera_neck, model.vtransform, model.fuser) +
                          ((model.head,) if model.head is not None else ())):
        synthetic.seed_modules_(m, 60 + i)
    return cfg, model.to(DEV).eval()


def test_camera_lidar_sweep_and_spatial_temporal_feature_selection(tmp_path):
    from al3d import sweep as S, synthetic
    from al3d.datasets import CameraLidarSweepLoader, PoolFrames
    from al3d.selectors import build_selector
    cfg, model = _build("bevfusion_camera_lidar_spatial_temporal_feature.py")
    assert cfg.model.type == "BEVFusion" and cfg.model.lidar.bbox_head is None and model.bbox_head is None
    n = 6
    pool = PoolFrames.from_synthetic(n, DEV, num_base=3, seed=11)

    def run(batch):
        loader = CameraLidarSweepLoader(pool, cfg.voxel_generator, None, batch, device=DEV, num_image_base=3, seed=5)
        return S.sweep_embeddings(model, loader, DEV, n), loader
    ref, loader = run(2)
    assert ref.shape == (n, 512) and bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    assert torch.equal(run(3)[0], ref) and torch.equal(run(1)[0], ref)             # batch-size invariant, deterministic
    saved = S.PIPELINE
    try:
        S.PIPELINE = None
        assert torch.equal(run(2)[0], ref)                                            # serial == rulebook-ahead pipeline
    finally:
        S.PIPELINE = saved
    # frames 0 and 3 share the lidar base cloud's shape class but not the cloud, frames 0 / 3 share the IMAGE base (3 bases):
    # every pair of frames must differ; and a camera-only change moves the embedding
    assert len({tuple(r.tolist()) for r in ref.cpu()}) == n
    ex = next(iter(loader))
    with torch.no_grad():
        out, middle = model(ex, return_loss=False, estimate=True)
        ex2 = dict(ex)
        ex2["img"] = ex["img"].flip(3)
        _, middle2 = model(ex2, return_loss=False, estimate=True)
    assert len(out) == 2 and out[0]["metadata"]["index"] == 0 and tuple(middle[-1].shape) == (2, 512, 180, 180)
    e1, e2 = middle[-1].mean(-1).mean(-1), middle2[-1].mean(-1).mean(-1)
    assert torch.equal(e1, ref[:2]) and not torch.equal(e1, e2)
    with pytest.raises(KeyError):
        model({k: v for k, v in ex.items() if k != "img"}, return_loss=False, estimate=True)
    # the selector on top
    infos, logs = synthetic.make_pool(1, seed=0)
    infos = infos[:n]
    ip, bp, lp = str(tmp_path / "infos.pkl"), str(tmp_path / "buffer.json"), str(tmp_path / "log.json")
    pickle.dump(infos, open(ip, "wb"))
    json.dump({"0": []}, open(bp, "w"))
    json.dump(logs, open(lp, "w"))
    loader.sampler = list(range(n))
    random.seed(3407)
    sel = build_selector(dict(type="SpatialTemporalFeatureSelector", budget=3, buffer_file=bp, infos_origin=ip, logs_file=lp,
                              detector=model, dataloader=loader, pred=True, buffer_path=str(tmp_path / "feat.pt"),
                              distance_store_file=None))
    sel.select_samples(local_rank=0)
    picked = sel.get_selected_samples()[sel.current_budget]
    assert 1 <= len(picked) <= n and len(set(picked)) == len(picked)
    assert torch.equal(torch.load(str(tmp_path / "feat.pt"), weights_only=True), ref.cpu())


def test_camera_lidar_detector_with_head_under_the_entropy_selector():
    from al3d import sweep as S
    from al3d.datasets import CameraLidarSweepLoader, PoolFrames
    cfg, model = _build("bevfusion_camera_lidar_entropy.py")
    assert type(model.bbox_head).__name__ == "TransFusionHead" and model.bbox_head.transpose_input
    n = 4
    pool = PoolFrames.from_synthetic(n, DEV, num_base=2, seed=3)
    loader = CameraLidarSweepLoader(pool, cfg.voxel_generator, None, 2, device=DEV, num_image_base=2, seed=1)
    emb, ent = S.sweep_embeddings(model, loader, DEV, n, with_entropy=True)
    assert emb.shape == (n, 512) and ent.shape == (n,) and bool(torch.isfinite(ent).all())
    with torch.no_grad():
        ref = []
        for ex in loader:
            for o in model(ex, return_loss=False, estimate=True)[0]:
                assert o["box3d_lidar"].shape[1] == 9 and 0 < len(o["scores"]) <= 200
                s = o["scores"].double()
                ref.append(float((-s * s.log() - (1 - s) * (1 - s).log()).mean()))
    np.testing.assert_allclose(ent.cpu().numpy(), np.asarray(ref), rtol=5e-6)
