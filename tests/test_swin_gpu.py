"""GPU suite: the Swin-T image backbone (al3d/models/swin.py; mmdet 2.20.0's ``SwinTransformer`` is not in the reference
tree -- parity unpinned, see the module docstring).  What can be pinned without the source:
  * the relative-position index built the mmdet way equals the published construction (coords_i - coords_j);
  * shifted-window attention (roll + window partition + additive region mask) equals a DENSE attention over all padded
    tokens with an independently derived "same window and same region" mask, for a map that needs padding;
  * output levels / strides / channels for the BEVFusion configuration, and the whole camera branch Swin-T ->
    GeneralizedLSSFPN -> DepthLSSTransform runs end to end on them (finite, right shapes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_relative_position_index_is_the_published_one():
    from al3d.models.swin import WindowMSA
    m = WindowMSA(96, 3, (7, 7))
    coords = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)      # [2, 49]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += 6
    rel[:, :, 1] += 6
    rel[:, :, 0] *= 13
    assert torch.equal(m.relative_position_index, rel.sum(-1))


def test_shifted_window_attention_equals_dense_masked_attention():
    from al3d.models.swin import ShiftWindowMSA
    torch.manual_seed(3)
    C, heads, ws, shift = 96, 3, 7, 3
    attn = ShiftWindowMSA(C, heads, ws, shift).to(DEV).eval()
    torch.nn.init.normal_(attn.w_msa.relative_position_bias_table, std=0.5)
    H, W = 16, 23                                                   # padded to 21 x 28
    x = torch.randn(2, H * W, C, device=DEV)
    with torch.no_grad():
        got = attn(x, (H, W))
        # ---- dense restatement
        Hp, Wp = 21, 28
        xp = torch.zeros(2, Hp, Wp, C, device=DEV)
        xp[:, :H, :W] = x.view(2, H, W, C)
        hh, ww = torch.meshgrid(torch.arange(Hp, device=DEV), torch.arange(Wp, device=DEV), indexing="ij")
        hs, wsft = (hh - shift) % Hp, (ww - shift) % Wp              # where a token sits after the cyclic shift
        win = (hs // ws) * (Wp // ws) + (wsft // ws)

        def region(v, n):
            return (v >= n - ws).long() + (v >= n - shift).long()
        reg = region(hs, Hp) * 3 + region(wsft, Wp)
        win, reg = win.flatten(), reg.flatten()
        allowed = (win[:, None] == win[None, :]) & (reg[:, None] == reg[None, :])
        ih, iw = (hs % ws).flatten(), (wsft % ws).flatten()
        idx = (ih[:, None] - ih[None, :] + ws - 1) * (2 * ws - 1) + (iw[:, None] - iw[None, :] + ws - 1)
        same_win = win[:, None] == win[None, :]
        m = attn.w_msa
        qkv = m.qkv(xp.view(2, Hp * Wp, C)).reshape(2, Hp * Wp, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        logits = (q * m.scale) @ k.transpose(-2, -1)                                        # [2, heads, L, L]
        bias = m.relative_position_bias_table[idx.clamp(0, (2 * ws - 1) ** 2 - 1)].permute(2, 0, 1)      # [heads, L, L]
        logits = logits + bias.unsqueeze(0)
        logits = logits.masked_fill(~same_win, float("-inf"))
        logits = logits + torch.where(allowed, 0.0, -100.0).to(logits.dtype)                # the reference's additive mask
        out = m.proj((logits.softmax(-1) @ v).transpose(1, 2).reshape(2, Hp * Wp, C))
        ref = out.view(2, Hp, Wp, C)[:, :H, :W].reshape(2, H * W, C)
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_swin_t_levels_and_camera_branch_end_to_end():
    from al3d.models import DepthLSSTransform, GeneralizedLSSFPN
    from al3d.models.swin import SwinTransformer
    from test_camera_branch_gpu import _camera_setup, _seed_
    torch.manual_seed(5)
    image_size, feature_size = (256, 704), (32, 88)
    swin = SwinTransformer(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4,
                           qkv_bias=True, patch_norm=True, out_indices=[1, 2, 3]).to(DEV).eval()
    names = set(swin.state_dict())
    for key in ("patch_embed.projection.weight", "patch_embed.norm.weight", "stages.0.blocks.1.attn.w_msa.qkv.bias",
                "stages.2.blocks.5.attn.w_msa.relative_position_bias_table", "stages.1.blocks.0.ffn.layers.0.0.weight",
                "stages.1.blocks.0.ffn.layers.1.bias", "stages.0.downsample.reduction.weight", "stages.2.downsample.norm.bias",
                "norm1.weight", "norm3.bias"):
        assert key in names, key
    assert "norm0.weight" not in names and "stages.3.downsample.norm.weight" not in names
    B, N = 1, 6
    img = torch.randn(B * N, *image_size, 3, device=DEV)
    neck = _seed_(GeneralizedLSSFPN([192, 384, 768], 256, 3), 7).to(DEV)
    vt = _seed_(DepthLSSTransform(256, 80, image_size, feature_size, [-54.0, 54.0, 0.3], [-54.0, 54.0, 0.3],
                                  [-10.0, 10.0, 20.0], [1.0, 60.0, 0.5], downsample=2), 8).to(DEV)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, points = _camera_setup(B, N, 9, image_size)
    with torch.no_grad():
        feats = swin(img)
        assert [tuple(f.shape) for f in feats] == [(6, 32, 88, 192), (6, 16, 44, 384), (6, 8, 22, 768)]
        assert all(bool(torch.isfinite(f).all()) for f in feats)
        fpn = neck(list(feats))
        assert tuple(fpn[0].shape) == (6, 32, 88, 256)
        bev = vt(fpn[0].view(B, N, 32, 88, 256), [p.to(DEV) for p in points], lidar2image.to(DEV), K.to(DEV),
                 cam2lidar.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
    assert tuple(bev.shape) == (1, 180, 180, 80) and bool(torch.isfinite(bev).all()) and float(bev.abs().max()) > 0


def test_assembled_camera_lidar_model_runs_and_is_deterministic():
    """``BEVFusionCameraLidar`` (fusion_models/bevfusion.py:207-305 restated on this build's modules): one synthetic
    sample through camera encoder, lidar encoder, fuser, decoder, embedding tap and TransFusionHead; shapes, finiteness,
    run-to-run identical embedding, and the camera map's [x, y] -> [H=y, W=x] transposition (a camera-only change must
    move the embedding, a lidar-only change too)."""
    import os
    from al3d import synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames
    from al3d.models import build_detector
    from al3d.models.bevfusion_model import BEVFusionCameraLidar, transfusion_head_for
    from al3d.utils import Config
    from test_camera_branch_gpu import _camera_setup, _seed_
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py"))
    lidar = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(lidar, seed=0)
    model = BEVFusionCameraLidar(lidar, head=transfusion_head_for())
    for i, m in enumerate((model.camera_backbone, model.camera_neck, model.vtransform, model.fuser, model.head)):
        _seed_(m, 40 + i)
    model = model.to(DEV).eval()
    pool = PoolFrames.from_synthetic(2, DEV, num_base=2, seed=1)
    ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, None, 1, device=DEV)))
    K, cam2lidar, lidar2image, img_aug, lidar_aug, _ = _camera_setup(1, 6, 9, (256, 704))
    img = torch.randn(1, 6, 256, 704, 3, generator=torch.Generator().manual_seed(2)).to(DEV)
    mats = (lidar2image.to(DEV), K.to(DEV), cam2lidar.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
    with torch.no_grad():
        emb, dec, preds = model(ex, img, [pool.frames[0]], *mats)
        emb2, _, _ = model(ex, img, [pool.frames[0]], *mats)
        emb_cam, _, _ = model(ex, img.flip(3), [pool.frames[0]], *mats)           # mirrored images only
    assert tuple(emb.shape) == (1, 512) and tuple(dec.shape) == (1, 180, 180, 512) and bool(torch.isfinite(emb).all())
    assert torch.equal(emb, emb2)
    assert not torch.equal(emb, emb_cam)
    assert len(preds) == 1 and preds[0]["bboxes"].shape[1] == 9 and len(preds[0]["scores"]) > 0
