"""BEVFusion lidar branch (BASELINE configs[3]) on this build's det3d-shaped modules.

CPU: the checkpoint converter covers every parameter of the embedding model.  GPU: the model with
converted weights reproduces the (x,y,z)-ordered CPU restatement of the reference's forward pass
(oracle/bevfusion_lidar.py; parity unpinned -- mmcv/mmdet/spconv absent, no reference fixture), and
at the config's full size (0.075 m voxels, 1440 x 1440 x 41 grid, 160k-voxel cap) the sweep is
finite, batch-size invariant bit for bit and feeds the selector."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
CFG = os.path.join(ROOT, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py")
DEV = "cuda:0"


def _model():
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(CFG)
    assert cfg.model.bbox_head is None and cfg.voxel_generator.max_voxel_num == 160000
    return cfg, build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)


def test_converter_covers_every_parameter():
    import bevfusion_lidar as ref
    from al3d.models.bevfusion_compat import convert_lidar_state_dict
    _, model = _model()
    sd = ref.make_state_dict(seed=3)
    conv = convert_lidar_state_dict(sd)
    own = model.state_dict()
    missing = [k for k in own if k not in conv and not k.endswith("num_batches_tracked")]
    extra = [k for k in conv if k not in own]
    assert not missing and not extra, (missing[:5], extra[:5])
    for k, v in conv.items():
        assert tuple(v.shape) == tuple(own[k].shape), k
    model.load_state_dict(conv, strict=False)
    # axis permutations: the (1,1,3) output conv over z becomes (3,1,1); asymmetric taps move with it
    w = sd["encoders.lidar.backbone.conv_out.0.weight"]
    assert tuple(conv["backbone.middle_conv3.2.weight"].shape) == (3, 1, 1, 128, 128)
    assert torch.equal(conv["backbone.middle_conv3.2.weight"][2, 0, 0], w[0, 0, 2])
    w2 = sd["decoder.backbone.blocks.0.0.weight"]
    assert torch.equal(conv["neck.blocks.0.1.weight"][:, :, 0, 2], w2[:, :, 2, 0])
    with pytest.raises(KeyError):
        convert_lidar_state_dict({k: v for k, v in sd.items() if "conv_out.1.running_var" not in k})


@pytest.mark.gpu
def test_lidar_branch_matches_xyz_restatement():
    """Random points on a small (x,y,z) = 64 x 48 x 41 grid: voxelise on device, run the converted
    model, compare the BEV map (transposed: this build's H is y) and the embedding with the dense
    CPU emulation in the reference's own axis order.  Tolerance: 23 stacked fp32-class convs,
    O(1) activations -> 2e-3 of the map's scale."""
    import bevfusion_lidar as ref
    from al3d import detector_ops as D
    from al3d.models.bevfusion_compat import convert_lidar_state_dict, to_bevfusion_coords
    _, model = _model()
    sd = ref.make_state_dict(seed=5)
    model.load_state_dict(convert_lidar_state_dict(sd), strict=False)
    model = model.to(DEV).eval()
    rng = np.random.default_rng(0)
    X, Y, Z = 64, 48, 41
    vs = np.array([0.075, 0.075, 0.2])
    lo = np.array([-2.4, -1.8, -5.0])
    batch, pts = 2, []
    for b in range(batch):
        n = 6000
        p = np.concatenate([rng.uniform(lo, lo + vs * [X, Y, Z - 1], size=(n, 3)),
                            rng.uniform(0, 255, (n, 1)), rng.uniform(0, 0.45, (n, 1))], axis=1).astype(np.float32)
        p[:, 2] = np.minimum(p[:, 2], -1.0 + 0.5 * rng.standard_normal(n).astype(np.float32))   # a thin slab, like a scan
        pts.append(torch.from_numpy(np.clip(p, [lo[0], lo[1], -4.99, 0, 0], [lo[0] + vs[0] * X - 1e-3, lo[1] + vs[1] * Y - 1e-3, 2.99, 255, 1]).astype(np.float32)).to(DEV))
    vox = D.Voxelizer([lo[0], lo[1], -5.0, lo[0] + vs[0] * X, lo[1] + vs[1] * Y, 3.0], list(vs), 10, 20000,
                      max_batch=batch, device=DEV)
    off = torch.tensor([0] + list(np.cumsum([p.shape[0] for p in pts])), dtype=torch.int64, device=DEV)
    out = vox(torch.cat(pts), off)
    feats, coords = out["feat"], out["coords"]
    assert [int(g) for g in vox.grid_size] == [X, Y, Z - 1]
    with torch.no_grad():
        x, middle = model.backbone(feats, coords, batch, np.array([X, Y, Z - 1]))
        bev = model.neck(x)                                              # [B, H=y, W=x, 512]
    want = ref.forward(sd, feats.cpu().numpy(), to_bevfusion_coords(coords.cpu()).numpy(), batch, (X, Y, Z))
    got = bev.permute(0, 3, 2, 1).cpu()                                  # -> [B, 512, x, y]
    assert got.shape == want.shape == (batch, 512, X // 8, Y // 8)
    scale = float(want.abs().max())
    assert scale > 0.1
    torch.testing.assert_close(got, want, rtol=0, atol=2e-3 * scale)
    emb = D.gap_nhwc(bev).cpu()
    torch.testing.assert_close(emb, want.mean(dim=(2, 3)), rtol=0, atol=5e-4 * scale)


@pytest.mark.gpu
def test_full_size_sweep_and_selection(tmp_path):
    """voxelnet_0p075 at full size: 10-sweep ~250k-point frames on the 1440 x 1440 x 41 grid.  No oracle
    finishes at this size in seconds, so properties: [N,512] finite embeddings, identical bits for batch
    sizes 1, 2 and 3 and across the pipeline modes, and the embedding file drives the selector."""
    from al3d import sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames
    cfg, model = _model()
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    pool = PoolFrames.from_synthetic(6, DEV, num_base=3, seed=11)

    def run(batch):
        loader = DeviceSweepLoader(pool, cfg.voxel_generator, None, batch, device=DEV)
        return S.sweep_embeddings(model, loader, DEV, len(pool))
    ref = run(2)
    assert ref.shape == (6, 512) and torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    assert torch.equal(run(3), ref) and torch.equal(run(1), ref)
    saved = S.PIPELINE
    try:
        for mode in (None, "split"):
            S.PIPELINE = mode
            assert torch.equal(run(2), ref), mode
    finally:
        S.PIPELINE = saved
    # more voxels than the CBGS grid ever sees: the 160k cap must be what limits a frame, not 60k
    loader = DeviceSweepLoader(pool, cfg.voxel_generator, None, 1, device=DEV)
    ex = next(iter(loader))
    assert int(ex["num_voxels"][0]) > 60000
    assert list(ex["shape"][0]) == [1440, 1440, 40]


@pytest.mark.gpu
def test_entropy_selector_end_to_end_on_the_lidar_detector(tmp_path):
    """BASELINE configs[3] with its detection half: voxelnet_0p075 encoder -> SECOND/SECONDFPN -> TransFusionHead behind
    the det3d contract (``detector(example, return_loss=False, estimate=True)`` -> per-frame dicts with ``scores``), swept
    at full size under the EntropySelector (entropy_selector.py:50-86): the per-frame entropies equal the formula on the
    detector's own scores, a frame without detections gives NaN (the reference's mean over an empty tensor), and the
    selector ranks by them under the cost budget.  Seeded weights, synthetic frames (parity unpinned)."""
    import json
    import pickle
    import random
    from al3d import sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames
    from al3d.models import build_detector
    from al3d.selectors import build_selector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "examples", "active", "bevfusion_lidar_entropy.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    n = 6
    pool = PoolFrames.from_synthetic(n, DEV, num_base=3, seed=11)
    loader = DeviceSweepLoader(pool, cfg.voxel_generator, None, 2, device=DEV)
    # the detector contract, one batch
    with torch.no_grad():
        out, middle = model(next(iter(loader)), return_loss=False, estimate=True)
    assert len(out) == 2 and set(out[0]) >= {"box3d_lidar", "scores", "label_preds", "metadata"}
    assert out[0]["box3d_lidar"].shape[1] == 9 and 0 < len(out[0]["scores"]) <= 200
    assert tuple(middle[-1].shape) == (2, 512, 180, 180)
    emb, ent = S.sweep_embeddings(model, loader, DEV, n, with_entropy=True)
    assert emb.shape == (n, 512) and ent.shape == (n,) and bool(torch.isfinite(ent).all())
    with torch.no_grad():
        ref = []
        for ex in loader:
            for o in model(ex, return_loss=False, estimate=True)[0]:
                s = o["scores"].double()
                ref.append(float((-s * s.log() - (1 - s) * (1 - s).log()).mean()))
    np.testing.assert_allclose(ent.cpu().numpy(), np.asarray(ref), rtol=5e-6)
    # no detection survives the score filter -> every frame's entropy is NaN (mean of an empty tensor)
    model.bbox_head.bbox_coder["score_threshold"] = 2.0
    _, ent_nan = S.sweep_embeddings(model, loader, DEV, n, with_entropy=True)
    assert bool(torch.isnan(ent_nan).all())
    model.bbox_head.bbox_coder["score_threshold"] = 0.0
    # the selector on top (tools/active_select.py:152-163 call sequence)
    infos, _ = synthetic.make_pool(1, seed=0)
    infos = infos[:n]
    ip, bp = str(tmp_path / "infos.pkl"), str(tmp_path / "buffer.json")
    pickle.dump(infos, open(ip, "wb"))
    json.dump({"0": []}, open(bp, "w"))
    loader.sampler = list(range(n))
    random.seed(3407)
    sel = build_selector(dict(type="EntropySelector", budget=3, buffer_file=bp, infos_origin=ip, detector=model,
                              dataloader=loader, pred=True, buffer_path=str(tmp_path / "entropy.pt")))
    sel.select_samples(local_rank=0)
    picked = sel.get_selected_samples()[sel.current_budget]
    order = np.argsort(-ent.cpu().numpy(), kind="stable").tolist()
    assert len(picked) >= 1 and picked == order[:len(picked)]
    assert torch.equal(torch.load(str(tmp_path / "entropy.pt"), weights_only=True), ent.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["BadgeSelector", "UWESelector", "PPALSelector"])
def test_weighted_feature_selectors_run_on_the_lidar_detector(cls, tmp_path):
    """The embedding x uncertainty selectors (badge / uwe / ppal) on the BEVFusion lidar-only detector with its TransFusionHead:
    they read per-frame ``scores`` / ``label_preds`` from plain dict predictions (the sweep's fallbacks) and the head's
    ``class_names``; property checks on the picks (distinct, within the pool, cost within the budget)."""
    import json
    import pickle
    import random
    from al3d import synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames
    from al3d.models import build_detector
    from al3d.selectors import build_selector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "examples", "active", "bevfusion_lidar_entropy.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    n = 8
    pool = PoolFrames.from_synthetic(n, DEV, num_base=4, seed=11)
    loader = DeviceSweepLoader(pool, cfg.voxel_generator, None, 2, device=DEV)
    loader.sampler = list(range(n))
    infos, _ = synthetic.make_pool(1, seed=0)
    infos = infos[:n]
    ip, bp = str(tmp_path / "infos.pkl"), str(tmp_path / "buffer.json")
    pickle.dump(infos, open(ip, "wb"))
    json.dump({"0": [], "1": [1]}, open(bp, "w"))     # a previous round: these selectors need a non-empty buffer (reference quirk)
    # PPAL's candidate loop has no pool bound (reference quirk): keep its expanded budget below the 8-frame pool's cost
    kw = dict(type=cls, budget=1 if cls == "PPALSelector" else 5, buffer_file=bp, infos_origin=ip, detector=model, dataloader=loader, pred=True,
              distance_store_file=None)
    if cls in ("BadgeSelector", "UWESelector"):
        kw.update(weighted_feat_path=str(tmp_path / "wf.pt"))
    else:
        cw = str(tmp_path / "cw.json")
        json.dump({c: 1.0 + 0.1 * i for i, c in enumerate(model.bbox_head.class_names[0])}, open(cw, "w"))
        kw.update(feat_path=str(tmp_path / "pf.pt"), ent_path=str(tmp_path / "pe.pt"), class_weight_file=cw)
    random.seed(3407)
    sel = build_selector(kw)
    sel.select_samples(local_rank=0)
    picked = sel.get_selected_samples()[sel.current_budget]
    assert 2 <= len(picked) <= n and len(set(picked)) == len(picked) and all(0 <= i < n for i in picked) and 1 in picked
