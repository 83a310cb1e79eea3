"""TEST INFRASTRUCTURE ONLY.  Golden vectors for three BEVFusion model stages of configs[4], produced by the reference's own
torch modules run on the CPU in this container with seeded parameters:

  * ``DepthLSSTransform`` (bevfusion/mmdet3d/models/vtransforms/depth_lss.py:14-102 on base.py:21-262): the lidar depth
    image, the frustum geometry, dtransform + depthnet + depth softmax x context (``get_cam_feats``), the voxel indices and
    the in-range filter of ``bev_pool`` (base.py:129-160), and ``downsample``;
  * ``ConvFuser`` (fusers/conv.py:11-25);
  * ``TransformerDecoderLayer`` with ``PositionEmbeddingLearned`` and the file's own ``MultiheadAttention``
    (utils/transformer.py:14-112,114-493): one layer of the TransFusion query decoder;
  * ``TransFusionBBoxCoder.decode`` (bevfusion/mmdet3d/core/bbox/coders/transfusion_bbox_coder.py:39-123, filter=True).

The modules import mmcv / mmdet3d registries at module level (absent here: ordinary ModuleNotFoundErrors); content-free
stand-ins are registered first: ``force_fp32`` as a decorator that returns the function unchanged, registries whose decorator
returns the class unchanged, placeholder names for ``mmcv.cnn`` (used only by classes this script never instantiates), and
for the compiled op ``mmdet3d.ops.bev_pool`` a RECORDER that keeps its arguments and returns zeros (the op is a compiled
extension, not run here).  The sum that op performs is then evaluated in this script as a float64 ``index_add_`` over the
recorded (features, voxel index) pairs -- that one step is a statement of the op's semantics (out[b, :, z, x, y] += feature),
not a reference output; everything up to the op's inputs, and ``downsample`` applied to that sum, IS the reference's code.
Run in the build container only (needs /root/reference); writes tests/golden/bevfusion_depth_lss.npz,
bevfusion_conv_fuser.npz, bevfusion_decoder_layer.npz, bevfusion_transfusion_decode.npz.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.environ.get("AL3D_REFERENCE_ROOT", "/root/reference")
BEV = os.path.join(ROOT, "bevfusion")
RECORD = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _force_fp32(*a, **k):
    return lambda fn: fn


def _bev_pool_recorder(x, geom_feats, B, D, H, W):
    RECORD["bev_pool"] = (x.detach().clone(), geom_feats.detach().clone(), int(B), int(D), int(H), int(W))
    return torch.zeros(int(B), x.shape[1], int(D), int(H), int(W), dtype=x.dtype)


def import_reference():
    if not os.path.isdir(BEV):
        raise RuntimeError(f"reference tree not found at {BEV}")
    _mod("mmcv")
    _mod("mmcv.runner", force_fp32=_force_fp32)
    _mod("mmcv.cnn", ConvModule=None, build_conv_layer=None, kaiming_init=None)
    _pkg("mmdet3d", os.path.join(BEV, "mmdet3d"))
    _mod("mmdet3d.ops", bev_pool=_bev_pool_recorder)
    _pkg("mmdet3d.models", os.path.join(BEV, "mmdet3d", "models"))
    _mod("mmdet3d.models.builder", VTRANSFORMS=_Registry(), FUSERS=_Registry())
    _pkg("mmdet3d.models.vtransforms", os.path.join(BEV, "mmdet3d", "models", "vtransforms"))
    _pkg("mmdet3d.models.fusers", os.path.join(BEV, "mmdet3d", "models", "fusers"))
    _pkg("mmdet3d.models.utils", os.path.join(BEV, "mmdet3d", "models", "utils"))
    depth_lss = importlib.import_module("mmdet3d.models.vtransforms.depth_lss")
    fuser = importlib.import_module("mmdet3d.models.fusers.conv")
    transformer = importlib.import_module("mmdet3d.models.utils.transformer")
    _mod("mmdet")
    _mod("mmdet.core")
    _mod("mmdet.core.bbox", BaseBBoxCoder=type("BaseBBoxCoder", (), {}))
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=_Registry())
    _pkg("mmdet3d.core", os.path.join(BEV, "mmdet3d", "core"))
    _pkg("mmdet3d.core.bbox", os.path.join(BEV, "mmdet3d", "core", "bbox"))
    _pkg("mmdet3d.core.bbox.coders", os.path.join(BEV, "mmdet3d", "core", "bbox", "coders"))
    coder = importlib.import_module("mmdet3d.core.bbox.coders.transfusion_bbox_coder")
    return depth_lss, fuser, transformer, coder


def seed_(module, seed):
    """Seeded parameters away from their defaults: weights ~ N(0, 1 / fan_in), biases and BatchNorm statistics non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue                                             # dx / bx / nx / frustum: geometry constants
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
    return module.eval()


def state_arrays(module, prefix):
    return {f"{prefix}{k}": v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def camera_rig(B, N, image_size, seed):
    """N cameras looking around the vehicle (x right, y down, z forward), a scaled + shifted image augmentation, a small
    lidar rotation + shift; points scattered around the vehicle."""
    g = torch.Generator().manual_seed(seed)
    iH, iW = image_size
    K = torch.eye(4).repeat(B, N, 1, 1)
    K[..., 0, 0] = K[..., 1, 1] = 0.48 * iW
    K[..., 0, 2], K[..., 1, 2] = iW / 2.0, iH / 2.0
    cam2lidar = torch.eye(4).repeat(B, N, 1, 1)
    for n in range(N):
        yaw = 2 * np.pi * n / N + 0.1
        fwd = torch.tensor([np.cos(yaw), np.sin(yaw), 0.0])
        right = torch.tensor([np.sin(yaw), -np.cos(yaw), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        cam2lidar[:, n, :3, :3] = torch.stack([right, down, fwd], 1).float()
        cam2lidar[:, n, :3, 3] = torch.tensor([0.5 * np.cos(yaw), 0.5 * np.sin(yaw), 1.5]).float()
    lidar2image = K.matmul(torch.inverse(cam2lidar))
    img_aug = torch.eye(4).repeat(B, N, 1, 1)
    img_aug[..., 0, 0] = img_aug[..., 1, 1] = 0.9
    img_aug[..., 0, 3], img_aug[..., 1, 3] = 3.0, -2.0
    lidar_aug = torch.eye(4).repeat(B, 1, 1)
    a = 0.05
    lidar_aug[:, 0, 0], lidar_aug[:, 0, 1], lidar_aug[:, 1, 0], lidar_aug[:, 1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    lidar_aug[:, :3, 3] = torch.tensor([0.3, -0.2, 0.05])
    points = [torch.cat([(torch.rand(3000, 2, generator=g) - 0.5) * 90.0, torch.rand(3000, 1, generator=g) * 4.0 - 2.0,
                         torch.rand(3000, 2, generator=g)], 1) for _ in range(B)]
    return K, cam2lidar, lidar2image, img_aug, lidar_aug, points


def gen_depth_lss(depth_lss, out):
    image_size, feature_size = (64, 176), (8, 22)
    B, N, Cin, C = 1, 3, 32, 16
    cfg = dict(in_channels=Cin, out_channels=C, image_size=image_size, feature_size=feature_size, xbound=[-54.0, 54.0, 1.2],
               ybound=[-54.0, 54.0, 1.2], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 31.0, 1.0], downsample=2)
    vt = seed_(depth_lss.DepthLSSTransform(**cfg), 41)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, points = camera_rig(B, N, image_size, 7)
    img = torch.randn(B, N, Cin, *feature_size, generator=torch.Generator().manual_seed(8))
    seen = {}
    cam_feats = vt.get_cam_feats

    def recording_get_cam_feats(x, d):                       # keeps the depth image the forward built and the features
        seen["depth"] = d.detach().clone()
        seen["feats"] = cam_feats(x, d)
        return seen["feats"]
    vt.get_cam_feats = recording_get_cam_feats
    eye = torch.eye(4).repeat(B, N, 1, 1)
    with torch.no_grad():
        vt(img, [p.clone() for p in points], eye, torch.eye(4).repeat(B, 1, 1), torch.inverse(cam2lidar), lidar2image, K,
           cam2lidar, img_aug, lidar_aug, None)
        geom = vt.get_geometry(cam2lidar[..., :3, :3], cam2lidar[..., :3, 3], K[..., :3, :3], img_aug[..., :3, :3],
                               img_aug[..., :3, 3], extra_rots=lidar_aug[..., :3, :3], extra_trans=lidar_aug[..., :3, 3])
        x_kept, idx_kept, b_, nz, nx, ny = RECORD["bev_pool"]
        # the op's sum (see the module docstring): out[b, :, z, x, y] += feature, in float64
        flat = ((idx_kept[:, 3] * nz + idx_kept[:, 2]) * nx + idx_kept[:, 0]) * ny + idx_kept[:, 1]
        pooled = torch.zeros(b_ * nz * nx * ny, x_kept.shape[1], dtype=torch.float64).index_add_(0, flat, x_kept.double())
        pooled = pooled.view(b_, nz, nx, ny, -1).permute(0, 4, 1, 2, 3).float()
        final = torch.cat(pooled.unbind(dim=2), 1)            # base.py:158: collapse Z
        down = vt.downsample(final)
    store = dict(cfg_image_size=np.array(image_size), cfg_feature_size=np.array(feature_size), cfg_channels=np.array([Cin, C]),
                 cfg_xbound=np.array(cfg["xbound"]), cfg_ybound=np.array(cfg["ybound"]), cfg_zbound=np.array(cfg["zbound"]),
                 cfg_dbound=np.array(cfg["dbound"]), img=img.numpy(), points=torch.stack(points).numpy(), K=K.numpy(),
                 cam2lidar=cam2lidar.numpy(), lidar2image=lidar2image.numpy(), img_aug=img_aug.numpy(), lidar_aug=lidar_aug.numpy(),
                 out_depth_image=seen["depth"].numpy(), out_geometry=geom.numpy(), out_cam_feats=seen["feats"].numpy(),
                 out_voxel_index=idx_kept.numpy().astype(np.int32), out_kept=np.array([x_kept.shape[0]]),
                 out_pooled=final.numpy(), out_downsample=down.numpy())
    store.update(state_arrays(vt, "sd."))
    np.savez_compressed(out, **store)
    print("wrote", out, "kept", x_kept.shape[0], "of", B * N * seen["feats"].shape[2] * feature_size[0] * feature_size[1],
          "pooled", tuple(final.shape), "down", tuple(down.shape))


def gen_conv_fuser(fuser, out):
    m = seed_(fuser.ConvFuser([16, 32], 32), 43)
    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(2, 16, 20, 24, generator=g), torch.randn(2, 32, 20, 24, generator=g)
    with torch.no_grad():
        y = m([a, b])
    store = dict(a=a.numpy(), b=b.numpy(), out=y.numpy())
    store.update(state_arrays(m, "sd."))
    np.savez_compressed(out, **store)
    print("wrote", out, tuple(y.shape))


def gen_decoder_layer(tr, out):
    C, heads, ffn, B, Pq, gh, gw = 128, 8, 256, 2, 50, 20, 20
    layer = tr.TransformerDecoderLayer(C, heads, ffn, dropout=0.1, activation="relu",
                                       self_posembed=tr.PositionEmbeddingLearned(2, C),
                                       cross_posembed=tr.PositionEmbeddingLearned(2, C))
    layer = seed_(layer, 47)
    g = torch.Generator().manual_seed(10)
    query = torch.randn(B, C, Pq, generator=g)
    key = torch.randn(B, C, gh * gw, generator=g)
    yy, xx = torch.meshgrid(torch.arange(gh, dtype=torch.float32) + 0.5, torch.arange(gw, dtype=torch.float32) + 0.5, indexing="ij")
    key_pos = torch.stack([xx, yy], -1).view(1, -1, 2).repeat(B, 1, 1)          # the BEV grid, shared by the samples
    query_pos = torch.rand(B, Pq, 2, generator=g) * torch.tensor([gw, gh])
    with torch.no_grad():
        y = layer(query, key, query_pos, key_pos)
    store = dict(query=query.numpy(), key=key.numpy(), query_pos=query_pos.numpy(), key_pos=key_pos.numpy(), out=y.numpy(),
                 cfg=np.array([C, heads, ffn]))
    store.update(state_arrays(layer, "sd."))
    np.savez_compressed(out, **store)
    print("wrote", out, tuple(y.shape))


def gen_decode(coder_mod, out):
    cfg = dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
               post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.05, code_size=10)
    coder = coder_mod.TransFusionBBoxCoder(**cfg)
    g = torch.Generator().manual_seed(12)
    B, ncls, P = 2, 10, 200
    heat = torch.rand(B, ncls, P, generator=g) ** 4                      # most scores small, some above the threshold
    rot = torch.randn(B, 2, P, generator=g)
    dim = torch.randn(B, 3, P, generator=g) * 0.5
    center = torch.rand(B, 2, P, generator=g) * 190.0 - 5.0              # feature-map cells, some outside the range
    height = torch.randn(B, 1, P, generator=g) * 4.0
    vel = torch.randn(B, 2, P, generator=g)
    store = dict(heat=heat.numpy(), rot=rot.numpy(), dim=dim.numpy(), center=center.numpy(), height=height.numpy(),
                 vel=vel.numpy(), score_threshold=np.array([cfg["score_threshold"]]))
    with torch.no_grad():
        res = coder.decode(heat.clone(), rot.clone(), dim.clone(), center.clone(), height.clone(), vel.clone(), filter=True)
    for i, r in enumerate(res):
        store[f"out{i}.bboxes"], store[f"out{i}.scores"], store[f"out{i}.labels"] = \
            r["bboxes"].numpy(), r["scores"].numpy(), r["labels"].numpy()
    np.savez_compressed(out, **store)
    print("wrote", out, [tuple(r["bboxes"].shape) for r in res])


def main():
    depth_lss, fuser, transformer, coder = import_reference()
    gold = os.path.join(os.path.dirname(HERE), "tests", "golden")
    gen_depth_lss(depth_lss, os.path.join(gold, "bevfusion_depth_lss.npz"))
    gen_conv_fuser(fuser, os.path.join(gold, "bevfusion_conv_fuser.npz"))
    gen_decoder_layer(transformer, os.path.join(gold, "bevfusion_decoder_layer.npz"))
    gen_decode(coder, os.path.join(gold, "bevfusion_transfusion_decode.npz"))


if __name__ == "__main__":
    main()
