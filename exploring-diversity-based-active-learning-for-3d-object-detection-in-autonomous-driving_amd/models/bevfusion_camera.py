"""BEVFusion camera branch, the part between the image encoder and the BEV decoder (BASELINE configs[4], SURVEY
section 8 row f4): Lift-Splat view transform (frustum geometry -> BEV pooling -> 2x downsample convs) and the
``ConvFuser`` that merges camera and lidar BEV maps.

Reference: bevfusion/mmdet3d/models/vtransforms/base.py:16-163 (``BaseTransform``: frustum, geometry, bev_pool),
vtransforms/depth_lss.py:58-102 (downsample, outer product), fusers/conv.py:11-25 (``ConvFuser``).  The image
backbone (Swin-T), the FPN neck and the depth net that produce ``depth`` / ``ctx`` are NOT built (mmcv / mmdet absent,
no checkpoints offline): this module takes their outputs.  Parameter names follow the reference
(``downsample.{0,1,3,4,6,7}``; fuser ``0``/``1``) so its state dicts load; maps are channels-last with this build's
[H=y, W=x] orientation handled by the caller as in ``bevfusion_compat`` (BEV maps here come out [B, nx0, nx1, C], i.e.
the reference's [H=x, W=y]).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from .. import detector_ops as D
from .. import lib
from ..selector_ops import _dev, _ptr, _stream


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


class _ConvBNReLU(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + ReLU on channels-last maps through the f16x3 / bf16x6 / f32 conv kernels."""

    def __init__(self, conv, bn):
        super().__init__()
        self.conv, self.bn = conv, bn
        self._packed = None

    def forward(self, x):
        key = (x.device, D.MATH, D.DENSE)
        if self._packed is None or self._packed[0] != key:
            scale, shift = D.fold_bn(self.bn)
            w, scale = D.pack_dense(D.pack_conv_weight(self.conv.weight).to(x.device), scale.to(x.device))
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
            points = kwargs["extra_rots"].view(B, 1, 1, 1, 1, 3, 3).repeat(1, N, 1, 1, 1, 1, 1) \
                .matmul(points.unsqueeze(-1)).squeeze(-1)
        if "extra_trans" in kwargs:
            points += kwargs["extra_trans"].view(B, 1, 1, 1, 1, 3).repeat(1, N, 1, 1, 1, 1)
        return points

    def geometry_device(self, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        """get_geometry with the per-point part on the device kernel (al3d_lss_geometry_f32): the 3x3 inverses and
        products per camera stay torch (B*N tiny matrices), the 2 M frustum points per sample do not."""
        B, N, _ = camera2lidar_trans.shape
        dev = self.frustum.device
        rows = torch.zeros((B * N, 44), dtype=torch.float32, device=dev)
        rows[:, 0:9] = torch.inverse(post_rots).reshape(B * N, 9)
        rows[:, 9:12] = post_trans.reshape(B * N, 3)
        rows[:, 12:21] = camera2lidar_rots.matmul(torch.inverse(intrins)).reshape(B * N, 9)
        rows[:, 21:24] = camera2lidar_trans.reshape(B * N, 3)
        if "extra_rots" in kwargs:
            rows[:, 24:33] = kwargs["extra_rots"].view(B, 1, 9).expand(B, N, 9).reshape(B * N, 9)
            rows[:, 36] = 1.0
        if "extra_trans" in kwargs:
            rows[:, 33:36] = kwargs["extra_trans"].view(B, 1, 3).expand(B, N, 3).reshape(B * N, 3)
            rows[:, 37] = 1.0
        Dd, fH, fW, _ = self.frustum.shape
        geom = torch.empty((B, N, Dd, fH, fW, 3), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.load().al3d_lss_geometry_workspace_bytes(B * N), dtype=torch.uint8, device=dev)
        lib.call("al3d_lss_geometry_f32", _ptr(self.frustum.contiguous()), Dd * fH * fW, _ptr(rows), B * N, _ptr(geom),
                 _ptr(ws), _stream())
        return geom

    def forward(self, depth, ctx, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        B, N, Dd, fH, fW = depth.shape
        geom = self.geometry_device(camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs)
        x = bev_pool(ctx.reshape(B * N, fH, fW, self.C).contiguous(), geom, B, self.dx.cpu().numpy(),
                     self.bx.cpu().numpy(), self.nx.cpu().numpy(), depth=depth.reshape(B * N, Dd, fH, fW).contiguous())
        for layer in self._ds:
            x = layer(x)
        return x


class ConvFuser(nn.Sequential):
    """fusers/conv.py:11-25: Conv2d(sum(in_channels), out_channels, 3, padding=1, bias=False) + BN + ReLU on the
    channel-concatenated BEV maps (channels-last here)."""

    def __init__(self, in_channels, out_channels):
        self.in_channels, self.out_channels = in_channels, out_channels
        super().__init__(nn.Conv2d(sum(in_channels), out_channels, 3, padding=1, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU(True))
        self._run = _ConvBNReLU(self[0], self[1])

    def forward(self, inputs):
        assert [t.shape[-1] for t in inputs] == list(self.in_channels)
        return self._run(torch.cat(inputs, dim=-1).contiguous())
