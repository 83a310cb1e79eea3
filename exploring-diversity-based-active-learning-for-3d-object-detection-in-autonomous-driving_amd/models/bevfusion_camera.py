"""BEVFusion camera branch, the part between the image encoder and the BEV decoder (BASELINE configs[4], SURVEY
section 8 row f4): Lift-Splat view transform (frustum geometry -> BEV pooling -> 2x downsample convs) and the
``ConvFuser`` that merges camera and lidar BEV maps.

Reference: bevfusion/mmdet3d/models/vtransforms/base.py:16-163 (``BaseTransform``: frustum, geometry, bev_pool),
vtransforms/depth_lss.py:14-102 (``DepthLSSTransform``: dtransform, depthnet, outer product, downsample),
necks/generalized_lss.py:13-110 (``GeneralizedLSSFPN``), fusers/conv.py:11-25 (``ConvFuser``).  The image backbone
(Swin-T) is NOT built (mmcv / mmdet absent, no checkpoints offline): the FPN takes its feature maps as input.  Parameter names follow the reference
(``downsample.{0,1,3,4,6,7}``; fuser ``0``/``1``) so its state dicts load; maps are channels-last with this build's
[H=y, W=x] orientation handled by the caller as in ``bevfusion_compat`` (BEV maps here come out [B, nx0, nx1, C], i.e.
the reference's [H=x, W=y]).
"""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from .. import detector_ops as D
from .. import lib
from ..selector_ops import _dev, _ptr, _stream
from .registry import NECKS


def gen_dx_bx(xbound, ybound, zbound):
    """vtransforms/base.py:16-22: float32 cell size, first cell centre, cell counts."""
    rows = [xbound, ybound, zbound]
    dx = torch.Tensor([row[2] for row in rows])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in rows])
    nx = torch.LongTensor([(row[1] - row[0]) / row[2] for row in rows])
    return dx, bx, nx


def bev_pool(x, geom, B, dx, bx, nx, depth=None, ctx_shape=None):
    """``BaseTransform.bev_pool`` (base.py:127-163).  x [P,C] f32 with geom [P,3] (materialised form), or -- with
    ``depth`` [BN,D,fH,fW] -- x = ctx [BN,fH,fW,C] and the outer product fused in.  -> [B, nx0, nx1, nx2*C]."""
    geom = _dev(geom.reshape(-1, 3), torch.float32, "geom")
    x = _dev(x, torch.float32, "x")
    dxn = np.asarray(dx, dtype=np.float32)
    lo = np.asarray(bx, dtype=np.float32) - dxn / np.float32(2.0)          # (bx - dx / 2.0) in float32
    nxn = np.asarray(nx, dtype=np.int32)
    C = x.shape[-1]
    P = geom.shape[0]
    out = torch.empty((B, int(nxn[0]), int(nxn[1]), int(nxn[2]) * C), dtype=torch.float32, device=x.device)
    ncell = B * int(nxn[0]) * int(nxn[1]) * int(nxn[2])
    ws = torch.empty(lib.load().al3d_bev_pool_workspace_bytes(P, ncell), dtype=torch.uint8, device=x.device)
    f3, i3 = ctypes.c_float * 3, ctypes.c_int * 3
    if depth is None:
        lib.call("al3d_bev_pool_f32", _ptr(x.reshape(-1, C)), _ptr(geom), P, C, B, f3(*lo.tolist()), f3(*dxn.tolist()),
                 i3(*nxn.tolist()), _ptr(out), _ptr(ws), _stream())
    else:
        depth = _dev(depth, torch.float32, "depth")
        BN, Dd, fH, fW = depth.shape
        assert tuple(x.shape) == (BN, fH, fW, C) and P == BN * Dd * fH * fW
        lib.call("al3d_bev_pool_lss_f32", _ptr(depth), _ptr(x), _ptr(geom), BN, Dd, fH, fW, C, B, f3(*lo.tolist()),
                 f3(*dxn.tolist()), i3(*nxn.tolist()), _ptr(out), _ptr(ws), _stream())
    return out


def _mat3(a, b):
    """[..., 3, 3] x [..., 3, 3] written out (a handful of per-camera matrices: no library GEMM launch for them)."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def _versions(*mods):
    """Version counters of every parameter / buffer of ``mods``: ``load_state_dict`` and in-place edits bump them, so a
    packed-weight cache keyed on this is rebuilt after weights change."""
    out = []
    for m in mods:
        if m is None:
            continue
        for t in list(m.parameters(recurse=False)) + list(m.buffers(recurse=False)):
            out.append((t.data_ptr(), t._version))
    return tuple(out)


class _ConvBNReLU(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + ReLU on channels-last maps through the f16x3 / bf16x6 / f32 conv kernels."""

    def __init__(self, conv, bn):
        super().__init__()
        self.conv, self.bn = conv, bn
        self._packed = None

    def forward(self, x):
        key = (x.device, D.MATH, D.DENSE, _versions(self.conv, self.bn))
        if self._packed is None or self._packed[0] != key:
            scale, shift = D.fold_bn(self.bn)
            c = self.conv
            geom = (c.kernel_size[0], c.stride[0], c.padding[0]) if c.in_channels % 16 == 0 else (None, None, None)
            w, scale = D.pack_dense(D.pack_conv_weight(c.weight).to(x.device), scale.to(x.device), *geom)
            self._packed = (key, w, scale, shift.to(x.device))
        _, w, scale, shift = self._packed
        return D.conv2d_nhwc(x, w, scale, shift, self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0], True)


class LSSViewTransform(nn.Module):
    """``DepthLSSTransform`` minus its depth net: frustum + geometry + BEV pooling + ``downsample``.

    forward(depth [B,N,D,fH,fW] softmax probabilities, ctx [B,N,fH,fW,C] context features (channels-last),
    camera2lidar_rots/trans, intrins, post_rots/trans [, extra_rots, extra_trans]) -> BEV map [B, nx0/ds, nx1/ds, C]."""

    def __init__(self, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound, downsample=1):
        super().__init__()
        self.image_size, self.feature_size = image_size, feature_size
        self.dbound = dbound
        dx, bx, nx = gen_dx_bx(xbound, ybound, zbound)
        self.dx, self.bx, self.nx = nn.Parameter(dx, requires_grad=False), nn.Parameter(bx, requires_grad=False), \
            nn.Parameter(nx, requires_grad=False)
        self.C = out_channels
        self.frustum = nn.Parameter(self.create_frustum(), requires_grad=False)
        self.D = self.frustum.shape[0]
        if downsample > 1:
            assert downsample == 2, downsample
            c = out_channels
            self.downsample = nn.Sequential(
                nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU(True),
                nn.Conv2d(c, c, 3, stride=downsample, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU(True),
                nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU(True))
            self._ds = [_ConvBNReLU(self.downsample[i], self.downsample[i + 1]) for i in (0, 3, 6)]
        else:
            self.downsample, self._ds = nn.Identity(), []

    def grid_numpy(self):
        """(dx, bx, nx) as numpy arrays, copied off the device once (they are constants of the configuration)."""
        key = (self.dx._version, self.bx._version, self.nx._version, self.dx.data_ptr())
        if getattr(self, "_grid_np", None) is None or self._grid_np[0] != key:
            self._grid_np = (key, self.dx.detach().cpu().numpy(), self.bx.detach().cpu().numpy(), self.nx.detach().cpu().numpy())
        return self._grid_np[1:]

    def create_frustum(self):
        """base.py:56-77."""
        iH, iW = self.image_size
        fH, fW = self.feature_size
        ds = torch.arange(*self.dbound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        Dd = ds.shape[0]
        xs = torch.linspace(0, iW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(Dd, fH, fW)
        ys = torch.linspace(0, iH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(Dd, fH, fW)
        return torch.stack((xs, ys, ds), -1)

    def get_geometry(self, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        """base.py:79-122 (small per-camera matrix algebra on the frustum: torch, on the maps' device)."""
        B, N, _ = camera2lidar_trans.shape
        points = self.frustum - post_trans.view(B, N, 1, 1, 1, 3)
        points = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(points.unsqueeze(-1))
        points = torch.cat((points[:, :, :, :, :, :2] * points[:, :, :, :, :, 2:3], points[:, :, :, :, :, 2:3]), 5)
        combine = camera2lidar_rots.matmul(torch.inverse(intrins))
        points = combine.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
        points += camera2lidar_trans.view(B, N, 1, 1, 1, 3)
        if "extra_rots" in kwargs:
            points = kwargs["extra_rots"].reshape(B, 1, 1, 1, 1, 3, 3).repeat(1, N, 1, 1, 1, 1, 1) \
                .matmul(points.unsqueeze(-1)).squeeze(-1)
        if "extra_trans" in kwargs:
            points += kwargs["extra_trans"].reshape(B, 1, 1, 1, 1, 3).repeat(1, N, 1, 1, 1, 1)
        return points

    def geometry_device(self, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        """get_geometry with the per-point part on the device kernel (al3d_lss_geometry_f32): the 3x3 inverses and
        products per camera stay torch (B*N tiny matrices), the 2 M frustum points per sample do not."""
        B, N, _ = camera2lidar_trans.shape
        return self._geometry_of_rows(self.geometry_rows(camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans,
                                                         **kwargs), B, N)

    def geometry_rows(self, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        """The 44 floats per camera al3d_lss_geometry_f32 reads (inverse post-rotation, post-translation, camera2lidar x
        inverse intrinsics, camera translation, the lidar augmentation): a function of the calibration matrices only."""
        B, N, _ = camera2lidar_trans.shape
        dev = self.frustum.device
        f = lambda t, k: t.reshape(B * N, k).to(dev, torch.float32)
        zeros = torch.zeros((B * N, 9), dtype=torch.float32, device=dev)
        has_rot, has_trans = "extra_rots" in kwargs, "extra_trans" in kwargs
        # one row of 44 floats per camera, assembled with ONE concatenation (slice assignments are a launch each)
        rows = torch.cat([
            f(torch.inverse(post_rots), 9), f(post_trans, 3),
            f(_mat3(camera2lidar_rots, torch.inverse(intrins)), 9), f(camera2lidar_trans, 3),
            f(kwargs["extra_rots"].reshape(B, 1, 9).expand(B, N, 9), 9) if has_rot else zeros,
            f(kwargs["extra_trans"].reshape(B, 1, 3).expand(B, N, 3), 3) if has_trans else zeros[:, :3],
            torch.full((B * N, 1), 1.0 if has_rot else 0.0, dtype=torch.float32, device=dev),
            torch.full((B * N, 1), 1.0 if has_trans else 0.0, dtype=torch.float32, device=dev),
            zeros[:, :6]], dim=1).contiguous()
        return rows

    def _geometry_of_rows(self, rows, B, N):
        dev = rows.device
        Dd, fH, fW, _ = self.frustum.shape
        geom = torch.empty((B, N, Dd, fH, fW, 3), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.load().al3d_lss_geometry_workspace_bytes(B * N), dtype=torch.uint8, device=dev)
        lib.call("al3d_lss_geometry_f32", _ptr(self.frustum.contiguous()), Dd * fH * fW, _ptr(rows), B * N, _ptr(geom),
                 _ptr(ws), _stream())
        return geom

    # The frustum geometry and the pooling plan (every frustum point's BEV cell, every cell's members in point order) are
    # functions of the calibration matrices only.  A sweep over a fixed camera rig hands the same matrices with every batch:
    # the last plan is kept and reused.  Whether the matrices are the same is decided ON THE HOST when the loader supplies
    # ``calib_key`` (a digest of the host bytes the matrices were uploaded from: no device round trip); without a key the
    # new rows are compared with the held ones on the device, which synchronises the stream once per batch.
    # The plan workspace (3 x P x 4 B + the cell arrays: ~0.4 GB at 16 samples of 6 cameras) stays resident with the
    # module and is replaced when the batch shape or the calibration changes; its build is ordered before any later use on
    # another stream by an event.  AL3D_LSS_PLAN_CACHE=0 rebuilds every time.
    PLAN_CACHE = os.environ.get("AL3D_LSS_PLAN_CACHE", "1") != "0"

    def pool_lss(self, depth, ctx, rows, B, N, calib_key=None):
        """Lift-Splat pooling of (depth [BN,D,fH,fW], ctx [BN,fH,fW,C]) under the cameras ``rows`` [BN,44]
        (geometry_rows): -> [B, nx0, nx1, nx2 * C].  ``calib_key``: host-side digest of the calibration (see above)."""
        dx, bx, nx = self.grid_numpy()
        dxn = np.asarray(dx, dtype=np.float32)
        lo = np.asarray(bx, dtype=np.float32) - dxn / np.float32(2.0)          # (bx - dx / 2.0) in float32
        nxn = np.asarray(nx, dtype=np.int32)
        depth, ctx = _dev(depth, torch.float32, "depth"), _dev(ctx, torch.float32, "ctx")
        BN, Dd, fH, fW = depth.shape
        C = ctx.shape[-1]
        P = BN * Dd * fH * fW
        f3, i3 = ctypes.c_float * 3, ctypes.c_int * 3
        held = getattr(self, "_plan", None)
        hit = self.PLAN_CACHE and held is not None and held[0].device == rows.device and held[3] == (B, P)
        if hit and calib_key is not None:
            hit = held[4] == calib_key                                  # host decision: no device round trip
        elif hit:
            hit = held[0].shape == rows.shape and bool(torch.equal(held[0], rows))     # synchronises (no key supplied)
        if not hit:
            geom = self._geometry_of_rows(rows, B, N)
            ncell = B * int(nxn[0]) * int(nxn[1]) * int(nxn[2])
            ws = torch.empty(lib.load().al3d_bev_pool_workspace_bytes(P, ncell), dtype=torch.uint8, device=depth.device)
            lib.call("al3d_bev_pool_plan", _ptr(geom.reshape(-1, 3)), P, B, f3(*lo.tolist()), f3(*dxn.tolist()), i3(*nxn.tolist()),
                     _ptr(ws), _stream())
            built = torch.cuda.Event()
            built.record(torch.cuda.current_stream(depth.device))
            held = (rows.clone(), ws, built, (B, P), calib_key)
            object.__setattr__(self, "_plan", held)
            self.plan_builds = getattr(self, "plan_builds", 0) + 1
        else:
            torch.cuda.current_stream(depth.device).wait_event(held[2])   # a plan built on another stream is complete
        out = torch.empty((B, int(nxn[0]), int(nxn[1]), int(nxn[2]) * C), dtype=torch.float32, device=depth.device)
        lib.call("al3d_bev_pool_lss_apply_f32", _ptr(depth), _ptr(ctx), BN, Dd, fH, fW, C, B, i3(*nxn.tolist()), _ptr(held[1]),
                 _ptr(out), _stream())
        return out

    def forward(self, depth, ctx, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        B, N, Dd, fH, fW = depth.shape
        rows = self.geometry_rows(camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs)
        x = self.pool_lss(depth.reshape(B * N, Dd, fH, fW).contiguous(), ctx.reshape(B * N, fH, fW, self.C).contiguous(), rows,
                          B, N)
        for layer in self._ds:
            x = layer(x)
        return x


def cat2_nhwc(a, b, a_hw_swapped=False):
    """cat([a, b], channel) of two channels-last maps as ONE device kernel (``al3d_cat2_nhwc_f32``); a may be stored
    [N, W, H, C] (``a_hw_swapped``)."""
    a, b = _dev(a, torch.float32, "a").contiguous(), _dev(b, torch.float32, "b").contiguous()
    N, H, W, Cb = b.shape
    want = (N, W, H) if a_hw_swapped else (N, H, W)
    if tuple(a.shape[:3]) != want:
        raise lib.Al3dError(f"cat2_nhwc: map sizes {tuple(a.shape)} / {tuple(b.shape)} do not match")
    out = torch.empty((N, H, W, a.shape[-1] + Cb), dtype=torch.float32, device=b.device)
    lib.call("al3d_cat2_nhwc_f32", _ptr(a), _ptr(b), N, H, W, a.shape[-1], Cb, 1 if a_hw_swapped else 0, _ptr(out), _stream())
    return out


class ConvFuser(nn.Sequential):
    """fusers/conv.py:11-25: Conv2d(sum(in_channels), out_channels, 3, padding=1, bias=False) + BN + ReLU on the
    channel-concatenated BEV maps (channels-last here)."""

    def __init__(self, in_channels, out_channels):
        self.in_channels, self.out_channels = in_channels, out_channels
        super().__init__(nn.Conv2d(sum(in_channels), out_channels, 3, padding=1, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU(True))
        # kept OUT of the module tree: the state dict holds the reference's keys (0.weight, 1.*) and nothing else
        object.__setattr__(self, "_run", _ConvBNReLU(self[0], self[1]))

    def forward(self, inputs, first_hw_swapped=False):
        """inputs: the channels-last BEV maps; ``first_hw_swapped``: the first one (the camera map) is still in the view
        transform's [x, y] order and is transposed by the concatenation kernel (no pass of its own)."""
        assert [t.shape[-1] for t in inputs] == list(self.in_channels)
        if len(inputs) == 2:
            return self._run(cat2_nhwc(inputs[0], inputs[1], first_hw_swapped))
        if first_hw_swapped:
            inputs = [inputs[0].permute(0, 2, 1, 3)] + list(inputs[1:])
        return self._run(torch.cat(inputs, dim=-1).contiguous())


class _ConvAffine(nn.Module):
    """Conv2d (+ bias) [+ BatchNorm2d] [+ ReLU] on channels-last maps through the dense conv kernels.  Input channels
    are zero-padded to a multiple of 16 (weights too), which the matrix-core kernels need; any Cout."""

    def __init__(self, conv, bn=None, relu=True):
        super().__init__()
        self.conv, self.bn, self.relu = conv, bn, relu
        self._packed = None

    def forward(self, x):
        key = (x.device, D.MATH, D.DENSE, _versions(self.conv, self.bn))
        conv = self.conv
        cin = conv.in_channels
        cpad = (cin + 15) // 16 * 16
        if self._packed is None or self._packed[0] != key:
            if self.bn is not None:
                scale, shift = D.fold_bn(self.bn)
                scale, shift = scale.cpu(), shift.cpu()
            else:
                scale = torch.ones(conv.out_channels)
                shift = torch.zeros(conv.out_channels)
            if conv.bias is not None:                                    # (x + b) * s + t
                shift = shift + conv.bias.detach().float().cpu() * scale
            w = conv.weight.detach().float()
            if cpad != cin:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cpad - cin))
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            wp, scale = D.pack_dense(D.pack_conv_weight(w).to(x.device), scale.to(x.device), k, s, p)
            self._packed = (key, wp, scale, shift.to(x.device))
        _, wp, scale, shift = self._packed
        if cpad != cin:
            x = torch.nn.functional.pad(x, (0, cpad - cin))
        return D.conv2d_nhwc(x.contiguous(), wp, scale, shift, conv.kernel_size[0], conv.stride[0], conv.padding[0],
                             self.relu)


class _ConvModule(nn.Module):
    """mmcv ``ConvModule`` (conv -> bn -> ReLU) with its parameter names (``conv.weight``, ``bn.*``)."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        object.__setattr__(self, "_run", _ConvAffine(self.conv, self.bn, True))     # not a submodule: no duplicate keys

    def forward(self, x):
        return self._run(x)


@NECKS.register_module
class GeneralizedLSSFPN(nn.Module):
    """bevfusion/mmdet3d/models/necks/generalized_lss.py:13-110: top-down path of upsample (bilinear,
    align_corners=True) -> concat -> 1x1 ConvModule -> 3x3 ConvModule per level.  Channels-last maps
    [BN, H_l, W_l, C_l] in, tuple of the ``used_backbone_levels`` outputs out.  Upsample + concatenation are one kernel
    (``al3d_lss_upsample_cat_f32``), the convolutions run on the dense conv kernels."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, upsample_cfg=None, **kwargs):
        super().__init__()
        # upsample_cfg: the class default is align_corners=True (generalized_lss.py:25); the shipped swint configs pass
        # {mode: bilinear, align_corners: false} (configs/.../camera+lidar/default.yaml:16-18) -- found by the round-5 golden
        upsample_cfg = dict(mode="bilinear", align_corners=True) if upsample_cfg is None else dict(upsample_cfg)
        if upsample_cfg.get("mode", "bilinear") != "bilinear":
            raise lib.Al3dError("GeneralizedLSSFPN: only bilinear upsampling is built")
        self.align_corners = bool(upsample_cfg.get("align_corners", False))
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.num_ins = len(in_channels)
        self.backbone_end_level = self.num_ins - 1 if end_level == -1 else end_level
        if end_level != -1:
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level = start_level
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            extra = in_channels[i + 1] if i == self.backbone_end_level - 1 else out_channels
            self.lateral_convs.append(_ConvModule(in_channels[i] + extra, out_channels, 1))
            self.fpn_convs.append(_ConvModule(out_channels, out_channels, 3, padding=1))

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [inputs[i + self.start_level] for i in range(len(inputs))]
        used = len(laterals) - 1
        for i in range(used - 1, -1, -1):
            lat, src = laterals[i].contiguous(), laterals[i + 1].contiguous()
            # upsample + concatenation as one kernel writing the channels-last concat map (al3d_lss_upsample_cat_f32): the
            # only backend -- a level whose channel count the kernel's float4 lanes cannot serve is an error, not a detour
            # through library ops
            if lat.shape[-1] % 4 or src.shape[-1] % 4:
                raise lib.Al3dError(f"GeneralizedLSSFPN: level channel counts {lat.shape[-1]} / {src.shape[-1]} must be "
                                    "multiples of 4 (al3d_lss_upsample_cat_f32)")
            x = torch.empty((*lat.shape[:3], lat.shape[-1] + src.shape[-1]), dtype=torch.float32, device=lat.device)
            lib.call("al3d_lss_upsample_cat_mode_f32", _ptr(_dev(lat, torch.float32, "lateral")), _ptr(_dev(src, torch.float32, "coarser level")),
                     lat.shape[0], lat.shape[1], lat.shape[2], lat.shape[3], src.shape[1], src.shape[2], src.shape[3],
                     1 if self.align_corners else 0, _ptr(x), _stream())
            laterals[i] = self.fpn_convs[i](self.lateral_convs[i](x))
        return tuple(laterals[i] for i in range(used))


class DepthLSSTransform(LSSViewTransform):
    """vtransforms/depth_lss.py:14-102 + ``BaseDepthTransform.forward`` (base.py:196-262): the depth-aware Lift-Splat
    transform of the camera+lidar configs.  The lidar points are rasterised into a per-camera depth image
    (``al3d_lss_depth_image_f32``), ``dtransform`` (1x1, 5x5/s4, 5x5/s2) brings it to the feature resolution, ``depthnet``
    (3x3, 3x3, 1x1 on [depth features | image features]) predicts D depth logits + C context channels, softmax over D,
    and the fused Lift-Splat pooling (``al3d_bev_pool_lss_f32``) never materialises the [B,N,D,fH,fW,C] product.

    forward(img [B,N,fH,fW,in_channels] channels-last image features (the FPN output), points (list of [P,>=3]),
    lidar2image [B,N,4,4], cam_intrinsic [B,N,4,4], camera2lidar [B,N,4,4], img_aug_matrix [B,N,4,4],
    lidar_aug_matrix [B,4,4]) -> BEV map [B, nx0/ds, nx1/ds, C]."""

    def __init__(self, in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound, downsample=1):
        super().__init__(out_channels, image_size, feature_size, xbound, ybound, zbound, dbound, downsample)
        self.in_channels = in_channels
        self.dtransform = nn.Sequential(
            nn.Conv2d(1, 8, 1), nn.BatchNorm2d(8), nn.ReLU(True),
            nn.Conv2d(8, 32, 5, stride=4, padding=2), nn.BatchNorm2d(32), nn.ReLU(True),
            nn.Conv2d(32, 64, 5, stride=2, padding=2), nn.BatchNorm2d(64), nn.ReLU(True))
        self.depthnet = nn.Sequential(
            nn.Conv2d(in_channels + 64, in_channels, 3, padding=1), nn.BatchNorm2d(in_channels), nn.ReLU(True),
            nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.BatchNorm2d(in_channels), nn.ReLU(True),
            nn.Conv2d(in_channels, self.D + self.C, 1))
        self._dt = [_ConvAffine(self.dtransform[i], self.dtransform[i + 1], True) for i in (0, 3, 6)]
        self._dn = [_ConvAffine(self.depthnet[0], self.depthnet[1], True), _ConvAffine(self.depthnet[3], self.depthnet[4], True),
                    _ConvAffine(self.depthnet[6], None, False)]

    def _calib_cached(self, slot, calib_key, build):
        """Small tensors that are functions of the calibration matrices only (matrix rows for the depth image, the frustum
        geometry rows): rebuilt with ``build()`` unless the loader's host-side digest says the calibration is the one they
        were built from -- a sweep over a fixed rig then launches none of the little inverse / concatenation kernels."""
        if calib_key is None:
            return build()
        cache = self.__dict__.setdefault("_calib_cache", {})
        held = cache.get(slot)
        if held is None or held[0] != calib_key:
            held = cache[slot] = (calib_key, build())
        return held[1]

    def depth_image(self, points, lidar2image, img_aug_matrix, lidar_aug_matrix, calib_key=None):
        """-> [B, N, iH, iW] f32: depth of the last lidar point (in point order) on every pixel, 0 where none."""
        B, N = lidar2image.shape[:2]
        iH, iW = self.image_size
        dev = self.frustum.device
        out = torch.empty((B, N, iH, iW), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.load().al3d_lss_depth_image_workspace_bytes(N, iH, iW), dtype=torch.uint8, device=dev)

        def build():
            # per-camera / per-sample matrices of the whole batch in two small tensors (one inverse, two concatenations)
            rows = torch.cat([lidar2image[..., :3, :3].reshape(B, N, 9), lidar2image[..., :3, 3],
                              img_aug_matrix[..., :3, :3].reshape(B, N, 9), img_aug_matrix[..., :3, 3]], dim=-1)
            aug = torch.cat([torch.inverse(lidar_aug_matrix[:, :3, :3]).reshape(B, 9), lidar_aug_matrix[:, :3, 3]], dim=-1)
            return rows.to(dev, torch.float32).contiguous(), aug.to(dev, torch.float32).contiguous()     # [B, N, 24], [B, 12]
        rows, aug = self._calib_cached("depth_rows", calib_key, build)
        for b in range(B):
            pts = _dev(points[b].float().contiguous(), torch.float32, "points")
            lib.call("al3d_lss_depth_image_f32", _ptr(pts), pts.shape[0], pts.shape[1], _ptr(rows[b]), N, _ptr(aug[b]), iH, iW,
                     _ptr(out[b]), _ptr(ws), _stream())
        return out

    def _dt01_fusable(self):
        c0, c1 = self.dtransform[0], self.dtransform[3]
        return (c0.in_channels, c0.out_channels, c0.kernel_size, c0.stride, c0.padding) == (1, 8, (1, 1), (1, 1), (0, 0)) and \
            (c1.in_channels, c1.out_channels, c1.kernel_size, c1.stride, c1.padding) == (8, 32, (5, 5), (4, 4), (2, 2))

    def _dtransform01(self, d):
        """dtransform[0:6] (1x1 conv + BN + ReLU -> 5x5 / stride 4 conv + BN + ReLU) as ONE kernel on the depth image
        [BN, iH, iW] -> [BN, oH, oW, 32] (``al3d_lss_dtransform01_f32``): the eight-channel full-resolution map -- 554 MB per
        16 samples, written, zero-padded to 16 channels and read back by the matrix-core path -- is never formed."""
        c0, b0, c1, b1 = self.dtransform[0], self.dtransform[1], self.dtransform[3], self.dtransform[4]
        key = (d.device, _versions(c0, b0, c1, b1))
        if getattr(self, "_dt01", None) is None or self._dt01[0] != key:
            def affine(conv, bn):
                scale, shift = D.fold_bn(bn)
                scale, shift = scale.cpu().float(), shift.cpu().float()
                if conv.bias is not None:
                    shift = shift + conv.bias.detach().float().cpu() * scale
                return scale, shift
            s0, t0 = affine(c0, b0)
            s1, t1 = affine(c1, b1)
            p0 = torch.cat([c0.weight.detach().float().cpu().reshape(8), s0, t0]).contiguous().to(d.device)
            w1 = c1.weight.detach().float().cpu().permute(2, 3, 1, 0).contiguous().to(d.device)      # [ky][kx][c][co]
            p1 = torch.cat([s1, t1]).contiguous().to(d.device)
            self._dt01 = (key, p0, w1, p1)
        _, p0, w1, p1 = self._dt01
        BN, iH, iW = d.shape
        d = _dev(d, torch.float32, "depth image")
        out = torch.empty((BN, (iH - 1) // 4 + 1, (iW - 1) // 4 + 1, 32), dtype=torch.float32, device=d.device)
        lib.call("al3d_lss_dtransform01_f32", _ptr(d), BN, iH, iW, _ptr(p0), _ptr(w1), _ptr(p1), _ptr(out), _stream())
        return out

    def get_cam_feats(self, x, d):
        """x [B,N,fH,fW,Cin], d [B,N,iH,iW] -> (depth probabilities [B*N,D,fH,fW], context [B*N,fH,fW,C])."""
        B, N, fH, fW, Cin = x.shape
        d = d.reshape(B * N, *d.shape[2:])
        if self._dt01_fusable():
            d = self._dt[2](self._dtransform01(d))
        else:
            d = d.unsqueeze(-1)
            for layer in self._dt:
                d = layer(d)
        assert d.shape[1:3] == (fH, fW), (d.shape, fH, fW)
        y = cat2_nhwc(d, x.reshape(B * N, fH, fW, Cin))
        for layer in self._dn:
            y = layer(y)
        y = y.contiguous()
        depth = torch.empty((B * N, self.D, fH, fW), dtype=torch.float32, device=y.device)
        lib.call("al3d_lss_depth_softmax_f32", _ptr(y), B * N, fH, fW, self.D, y.shape[-1], _ptr(depth), _stream())
        return depth, y[..., self.D:self.D + self.C].contiguous()

    def forward(self, img, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix, calib_key=None):
        B, N, fH, fW, _ = img.shape
        d = self.depth_image(points, lidar2image, img_aug_matrix, lidar_aug_matrix, calib_key=calib_key)
        depth, ctx = self.get_cam_feats(img, d)
        rows = self._calib_cached("geometry_rows", calib_key, lambda: self.geometry_rows(
            camera2lidar[..., :3, :3], camera2lidar[..., :3, 3], cam_intrinsic[..., :3, :3],
            img_aug_matrix[..., :3, :3], img_aug_matrix[..., :3, 3],
            extra_rots=lidar_aug_matrix[..., :3, :3], extra_trans=lidar_aug_matrix[..., :3, 3]))
        x = self.pool_lss(depth, ctx, rows, B, N, calib_key=calib_key)
        for layer in self._ds:
            x = layer(x)
        return x
