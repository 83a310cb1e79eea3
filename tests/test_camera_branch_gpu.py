"""GPU suite: the camera branch between the image backbone and the fused BEV map (BASELINE
configs[4]): ``GeneralizedLSSFPN`` and ``DepthLSSTransform`` (lidar depth image, dtransform, depthnet, fused
Lift-Splat pooling, downsample) against plain-torch restatements of the reference modules with the same weights.

Reference: bevfusion/mmdet3d/models/necks/generalized_lss.py:13-110, vtransforms/depth_lss.py:14-102,
vtransforms/base.py:196-262.  mmcv / mmdet are not importable here and no checkpoint exists offline, so these are
restatements with seeded weights (parity unpinned, like the rest of the BEVFusion rows); the tests of this file
start from synthetic backbone feature maps (the Swin-T backbone: tests/test_swin_gpu.py).

Tolerances (fp32-class f16x3 convolutions against torch fp32; stated per check)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


from al3d.synthetic import camera_setup as _camera_setup, seed_modules_ as _seed_  # noqa: E402


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def test_generalized_lss_fpn_matches_torch_restatement():
    from al3d.models.bevfusion_camera import GeneralizedLSSFPN
    neck = _seed_(GeneralizedLSSFPN([192, 384, 768], 256, 3), 1)
    assert sorted(neck.state_dict())[:3] == ["fpn_convs.0.bn.bias", "fpn_convs.0.bn.num_batches_tracked",
                                             "fpn_convs.0.bn.running_mean"]            # mmcv ConvModule names
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(6, c, h, w, generator=g) for c, h, w in ((192, 32, 88), (384, 16, 44), (768, 8, 22))]
    # restatement of generalized_lss.py:86-110 in NCHW torch
    lat = list(feats)
    with torch.no_grad():
        for i in (1, 0):
            up = F.interpolate(lat[i + 1], size=lat[i].shape[2:], mode="bilinear", align_corners=True)
            x = torch.cat([lat[i], up], 1)
            lc, fc = neck.lateral_convs[i], neck.fpn_convs[i]
            x = F.relu(lc.bn(lc.conv(x)))
            lat[i] = F.relu(fc.bn(fc.conv(x)))
        neck = neck.to(DEV)
        outs = neck([_nhwc(f).to(DEV) for f in feats])
    assert len(outs) == 2
    for i in (0, 1):
        ref = _nhwc(lat[i])
        got = outs[i].cpu()
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-5


def _depth_image_reference(points, lidar2image, img_aug_matrix, lidar_aug_matrix, image_size):
    """What BaseDepthTransform.forward (base.py:225-262) rasterises, evaluated independently in float64 numpy: undo the
    lidar augmentation, project with lidar2image, apply the image augmentation, keep points inside the image, and let
    the LAST point (in point order) that lands on a pixel set its depth."""
    iH, iW = image_size
    B, N = lidar2image.shape[:2]
    out = np.zeros((B, N, iH, iW), np.float32)
    for b in range(B):
        A = lidar_aug_matrix[b].double().numpy()
        xyz = points[b][:, :3].double().numpy()
        raw = (np.linalg.inv(A[:3, :3]) @ (xyz - A[:3, 3]).T)                 # [3, P] points before the lidar augmentation
        for c in range(N):
            M = lidar2image[b, c].double().numpy()
            G = img_aug_matrix[b, c].double().numpy()
            cam = M[:3, :3] @ raw + M[:3, 3:4]
            cam[2] = np.clip(cam[2], 1e-5, 1e5)
            depth = cam[2]                                                   # base.py:236-237: a view of the clamped row
            cam[:2] /= cam[2:3]
            pix = G[:3, :3] @ cam + G[:3, 3:4]
            col, row = pix[0], pix[1]
            inside = (row >= 0) & (row < iH) & (col >= 0) & (col < iW)
            for j in np.nonzero(inside)[0]:                                  # ascending point order: later points overwrite
                out[b, c, int(row[j]), int(col[j])] = depth[j]
    return torch.from_numpy(out)


def test_lidar_depth_image_matches_reference_loop():
    from al3d.models.bevfusion_camera import DepthLSSTransform
    image_size, feature_size = (64, 176), (8, 22)
    vt = DepthLSSTransform(32, 16, image_size, feature_size, [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-10.0, 10.0, 20.0],
                           [1.0, 60.0, 1.0], downsample=2).to(DEV)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, points = _camera_setup(2, 6, 3, image_size)
    ref = _depth_image_reference(points, lidar2image, img_aug, lidar_aug, image_size)
    got = vt.depth_image([p.to(DEV) for p in points], lidar2image.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV)).cpu()
    assert got.shape == ref.shape and float((ref > 0).float().mean()) > 0.01
    # same projection in another summation order: a point within float rounding of a pixel border may land in the
    # neighbouring pixel, and where two points share a pixel the surviving depth may then differ
    both = (ref > 0) & (got > 0)
    agree = both & ((ref - got).abs() <= 1e-3 * ref.abs())
    n_hit = int(((ref > 0) | (got > 0)).sum())
    assert int(agree.sum()) >= 0.998 * n_hit, (int(agree.sum()), n_hit)


def test_depth_image_stores_the_clamped_depth_for_points_behind_the_camera():
    """base.py:236-237: ``dist`` aliases the row that ``torch.clamp`` then overwrites, so a point at or behind the camera
    plane that still lands on a pixel is rasterised with depth 1e-5.  Identity projection: q = point."""
    from al3d.models.bevfusion_camera import DepthLSSTransform
    image_size, feature_size = (64, 176), (8, 22)
    vt = DepthLSSTransform(32, 16, image_size, feature_size, [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-10.0, 10.0, 20.0],
                           [1.0, 60.0, 1.0], downsample=2).to(DEV)
    eye = torch.eye(4).reshape(1, 1, 4, 4)
    pts = torch.tensor([[3.05e-4, 2.05e-4, -1.0, 0.0, 0.0],       # behind: z -> 1e-5, pixel (row 20, col 30)
                        [40.5, 10.5, 1.0, 0.0, 0.0],              # in front: depth 1, pixel (row 10, col 40)
                        [5.05e-4, 4.05e-4, 0.0, 0.0, 0.0]])       # on the plane: z -> 1e-5, pixel (row 40, col 50)
    ref = _depth_image_reference([pts], eye, eye, torch.eye(4).reshape(1, 4, 4), image_size)
    got = vt.depth_image([pts.to(DEV)], eye.to(DEV), eye.to(DEV), torch.eye(4).reshape(1, 4, 4).to(DEV)).cpu()
    got = got.reshape(ref.shape)
    assert ref[0, 0, 20, 30] == np.float32(1e-5) and ref[0, 0, 40, 50] == np.float32(1e-5) and ref[0, 0, 10, 40] == 1.0
    assert torch.equal(got, ref)


def test_depth_lss_transform_matches_torch_restatement():
    """Whole DepthLSSTransform.forward: depth image -> dtransform -> depthnet -> softmax x context -> BEV pooling
    -> downsample, against torch fp32 modules for the conv stacks and the materialised outer product pooled by
    this build's (separately tested) ``bev_pool``."""
    from al3d.models.bevfusion_camera import DepthLSSTransform, bev_pool
    image_size, feature_size = (64, 176), (8, 22)
    B, N, Cin, C = 1, 6, 32, 16
    vt = _seed_(DepthLSSTransform(Cin, C, image_size, feature_size, [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6],
                                  [-10.0, 10.0, 20.0], [1.0, 60.0, 1.0], downsample=2), 5)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, points = _camera_setup(B, N, 4, image_size)
    g = torch.Generator().manual_seed(6)
    img = torch.randn(B, N, Cin, *feature_size, generator=g)
    vt = vt.to(DEV)
    with torch.no_grad():
        d = vt.depth_image([p.to(DEV) for p in points], lidar2image.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
        # restatement of get_cam_feats (depth_lss.py:82-97) in NCHW torch on the same depth image
        dd = vt.dtransform(d.reshape(B * N, 1, *image_size))
        x = vt.depthnet(torch.cat([dd, img.reshape(B * N, Cin, *feature_size).to(DEV)], 1))
        depth_ref = x[:, :vt.D].softmax(dim=1)
        ctx_ref = x[:, vt.D:vt.D + C]
        depth, ctx = vt.get_cam_feats(img.permute(0, 1, 3, 4, 2).contiguous().to(DEV), d)
        assert float((depth - depth_ref).abs().max()) <= 2e-4            # probabilities
        assert float((ctx.permute(0, 3, 1, 2) - ctx_ref).abs().max()) <= 1e-4 * float(ctx_ref.abs().max()) + 1e-5
        got = vt(img.permute(0, 1, 3, 4, 2).contiguous().to(DEV), [p.to(DEV) for p in points], lidar2image.to(DEV),
                 K.to(DEV), cam2lidar.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
        # reference: materialised product -> bev_pool -> torch downsample
        geom = vt.get_geometry(cam2lidar[..., :3, :3].to(DEV), cam2lidar[..., :3, 3].to(DEV), K[..., :3, :3].to(DEV),
                               img_aug[..., :3, :3].to(DEV), img_aug[..., :3, 3].to(DEV),
                               extra_rots=lidar_aug[..., :3, :3].to(DEV), extra_trans=lidar_aug[..., :3, 3].to(DEV))
        prod = depth_ref.unsqueeze(1) * ctx_ref.unsqueeze(2)                         # [BN, C, D, fH, fW]
        prod = prod.view(B, N, C, vt.D, *feature_size).permute(0, 1, 3, 4, 5, 2)      # depth_lss.py:93-96
        bev = bev_pool(prod.reshape(-1, C).contiguous(), geom, B, vt.dx.cpu().numpy(), vt.bx.cpu().numpy(),
                       vt.nx.cpu().numpy())
        ref = vt.downsample(bev.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert got.shape == ref.shape == (B, 90, 90, C)
    assert float(ref.abs().max()) > 0
    assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-5


def test_depth_image_without_points_is_all_zero():
    from al3d.models.bevfusion_camera import DepthLSSTransform
    vt = DepthLSSTransform(32, 16, (64, 176), (8, 22), [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-10.0, 10.0, 20.0],
                           [1.0, 60.0, 1.0], downsample=2).to(DEV)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, _ = _camera_setup(1, 6, 3, (64, 176))
    d = vt.depth_image([torch.zeros((0, 5), device=DEV)], lidar2image.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
    assert d.shape == (1, 6, 64, 176) and float(d.abs().max()) == 0.0


@pytest.mark.parametrize("size", [(64, 176), (37, 53), (5, 4)])
def test_fused_dtransform_head_matches_float64(size):
    """``al3d_lss_dtransform01_f32`` (1x1 conv + BN + ReLU -> 5x5 / stride 4 / padding 2 conv + BN + ReLU as one fp32
    kernel, depth_lss.py:38-44) against the two torch layers evaluated in float64: the bound is fp32 rounding of a
    200-term sum, 1e-5 of the largest output; image sizes that are not multiples of the stride exercise the zero
    padding of the FIRST layer's output (a tap outside the image adds 0, not relu(shift0))."""
    from al3d.models.bevfusion_camera import DepthLSSTransform
    vt = _seed_(DepthLSSTransform(32, 16, (64, 176), (8, 22), [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-10.0, 10.0, 20.0],
                                  [1.0, 60.0, 1.0], downsample=2), 11)
    with torch.no_grad():
        vt.dtransform[0].bias.copy_(torch.linspace(-0.5, 0.8, 8))          # relu(shift0) != 0 on empty pixels
    vt = vt.to(DEV).eval()
    assert vt._dt01_fusable()
    g = torch.Generator().manual_seed(2)
    d = torch.rand(3, *size, generator=g) * 60.0
    d[torch.rand(3, *size, generator=g) < 0.7] = 0.0                        # lidar depth images are mostly empty
    with torch.no_grad():
        got = vt._dtransform01(d.to(DEV))
        ref = vt.dtransform[:6].double()(d.double().unsqueeze(1).to(DEV)).permute(0, 2, 3, 1)
        vt.dtransform.float()
    assert got.shape == ref.shape and got.shape[-1] == 32
    assert float(ref.abs().max()) > 0
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("swapped", [False, True])
def test_cat2_kernel_equals_torch_cat(swapped):
    """``al3d_cat2_nhwc_f32``: the channel concatenation in front of the depth net and of the fuser (depth_lss.py:84,
    fusers/conv.py:24) as one kernel, bit for bit == ``torch.cat`` -- with the camera map's [x, y] -> [y, x] transposition
    riding on the copy."""
    from al3d.models.bevfusion_camera import cat2_nhwc
    g = torch.Generator().manual_seed(5)
    N, H, W, Ca, Cb = 2, 9, 14, 80, 256
    a = torch.randn(N, W, H, Ca, generator=g).to(DEV) if swapped else torch.randn(N, H, W, Ca, generator=g).to(DEV)
    b = torch.randn(N, H, W, Cb, generator=g).to(DEV)
    want = torch.cat([a.permute(0, 2, 1, 3) if swapped else a, b], dim=-1)
    assert torch.equal(cat2_nhwc(a, b, swapped), want)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape", [((2, 32, 88, 192), (16, 44, 384)), ((1, 7, 5, 8), (3, 2, 12)), ((1, 4, 4, 4), (1, 1, 4))])
def test_upsample_cat_kernel_matches_torch(shape, align):
    """``al3d_lss_upsample_cat_f32`` (generalized_lss.py:88-101: bilinear upsample with align_corners=True of the coarser
    level + channel concatenation, one kernel on channels-last maps) against ``F.interpolate`` + ``torch.cat``: the
    copied half bit for bit, the interpolated half to fp32 rounding of a four-tap blend against torch's float32 result;
    odd sizes and a 1 x 1 source (every weight on one tap) included."""
    import ctypes
    from al3d import lib
    (N, H, W, C1), (h, w, C2) = shape
    g = torch.Generator().manual_seed(H * W + C2)
    lat = torch.randn(N, H, W, C1, generator=g).to(DEV)
    src = torch.randn(N, h, w, C2, generator=g).to(DEV)
    out = torch.empty(N, H, W, C1 + C2, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.call("al3d_lss_upsample_cat_mode_f32", p(lat), p(src), N, H, W, C1, h, w, C2, 1 if align else 0, p(out),
             torch.cuda.current_stream().cuda_stream)
    up64 = F.interpolate(src.permute(0, 3, 1, 2).double(), size=(H, W), mode="bilinear", align_corners=align).permute(0, 2, 3, 1)
    up32 = F.interpolate(src.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=align).permute(0, 2, 3, 1)
    scale = float(src.abs().max())
    assert torch.equal(out[..., :C1], lat)
    # the source position o * (in - 1) / (out - 1) is a float32 product in torch and here: against torch's own float32 result
    # only the blend's rounding remains; against exact positions the position's rounding (1 ulp of ~16) times a tap difference
    assert float((out[..., C1:] - up32).abs().max()) <= 2e-6 * scale
    assert float((out[..., C1:].double() - up64).abs().max()) <= 2e-5 * scale


def test_pooling_plan_is_cached_per_calibration_and_equals_the_one_shot_call():
    """The Lift-Splat plan (frustum geometry, every point's BEV cell, every cell's members in point order) depends on the
    calibration matrices only: plan + apply == the one-shot al3d_bev_pool_lss_f32 bit for bit; a second batch under the SAME
    rig reuses the plan (no new build) and still gives the one-shot result; changed matrices rebuild it."""
    from al3d.models.bevfusion_camera import DepthLSSTransform, bev_pool
    image_size, feature_size = (64, 176), (8, 22)
    B, N, C = 2, 6, 16
    vt = _seed_(DepthLSSTransform(32, C, image_size, feature_size, [-54.0, 54.0, 0.6], [-54.0, 54.0, 0.6], [-10.0, 10.0, 20.0],
                                  [1.0, 60.0, 1.0], downsample=2), 5).to(DEV)
    K, cam2lidar, _, img_aug, lidar_aug, _ = _camera_setup(B, N, 4, image_size)
    g = torch.Generator().manual_seed(8)

    def maps():
        depth = torch.rand(B * N, vt.D, *feature_size, generator=g).softmax(1).to(DEV)
        ctx = torch.randn(B * N, *feature_size, C, generator=g).to(DEV)
        return depth, ctx

    def rows(c2l):
        return vt.geometry_rows(c2l[..., :3, :3].to(DEV), c2l[..., :3, 3].to(DEV), K[..., :3, :3].to(DEV),
                                img_aug[..., :3, :3].to(DEV), img_aug[..., :3, 3].to(DEV),
                                extra_rots=lidar_aug[..., :3, :3].to(DEV), extra_trans=lidar_aug[..., :3, 3].to(DEV))

    def one_shot(depth, ctx, r):
        geom = vt._geometry_of_rows(r, B, N)
        return bev_pool(ctx, geom, B, *vt.grid_numpy(), depth=depth)
    vt.plan_builds = 0
    r1 = rows(cam2lidar)
    d1, c1 = maps()
    assert torch.equal(vt.pool_lss(d1, c1, r1, B, N), one_shot(d1, c1, r1)) and vt.plan_builds == 1
    d2, c2 = maps()
    assert torch.equal(vt.pool_lss(d2, c2, rows(cam2lidar), B, N), one_shot(d2, c2, r1)) and vt.plan_builds == 1   # same rig: reused
    moved = cam2lidar.clone()
    moved[..., 0, 3] += 0.25
    r3 = rows(moved)
    out3 = vt.pool_lss(d2, c2, r3, B, N)
    assert vt.plan_builds == 2 and torch.equal(out3, one_shot(d2, c2, r3)) and not torch.equal(out3, one_shot(d2, c2, r1))
    # with a host-side calibration digest the decision needs no device comparison: same key -> reuse, other
    # key -> rebuild, and a plan used from another stream waits for the event recorded at its build
    out4 = vt.pool_lss(d2, c2, r3, B, N, calib_key=b"rig-a")
    assert vt.plan_builds == 3 and torch.equal(out4, out3)             # the held plan carried no key: rebuilt once, keyed
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out5 = vt.pool_lss(d2, c2, r3, B, N, calib_key=b"rig-a")
    torch.cuda.current_stream().wait_stream(side)
    assert vt.plan_builds == 3 and torch.equal(out5, out3)
    out6 = vt.pool_lss(d1, c1, r1, B, N, calib_key=b"rig-b")
    assert vt.plan_builds == 4 and torch.equal(out6, one_shot(d1, c1, r1))
