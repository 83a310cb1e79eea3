"""CPU restatement of BEVFusion's lidar branch (BASELINE configs[3]): test infrastructure only.

Follows, in the reference's OWN axis order (coords (batch, x, y, z), BEV maps [N, C, H=x, W=y]) and
with the reference's OWN state-dict names / weight layouts:

  * hard voxelisation + mean reduce   bevfusion/mmdet3d/models/fusion_models/bevfusion.py:137-160,
                                      ops/voxel/src/voxelization_cuda.cu:106-180 (first-appearance
                                      order, first max_points points per voxel, first max_voxels)
  * SparseEncoder.forward             bevfusion/mmdet3d/models/backbones/sparse_encoder.py:104-220
                                      (config configs/nuscenes/det/transfusion/secfpn/lidar/voxelnet_0p075.yaml)
  * SparseBasicBlock                  bevfusion/mmdet3d/ops/sparse_block.py:62-110
  * SECOND / SECONDFPN                bevfusion/mmdet3d/models/backbones/second.py:28-95,
                                      necks/second.py:30-100 (configs/.../secfpn/default.yaml)

Sparse convolutions are emulated densely (spconv semantics from the vendored sources,
bevfusion/mmdet3d/ops/spconv/include/spconv/geometry.h:25-82): SubMConv3d = conv3d(pad k//2) restricted
to the active sites; SparseConv3d = strided conv3d of the zero-filled tensor, active outputs = sites
with an active input under the kernel; BN(eval) + ReLU on active sites only.

PARITY UNPINNED: mmcv / mmdet / torchpack / spconv are absent, so the reference itself cannot run
here and holds no fixture for this path; this file is a second, independent statement of the
documented forward pass that the GPU path (different axis order, sparse kernels, converted weights)
is checked against.  Only tests import it.
"""
import numpy as np
import torch
import torch.nn.functional as F

ENCODER_CHANNELS = ((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128))
DOWN_PADDING = (1, 1, (1, 1, 0))           # encoder_paddings, last entry of stages 1..3 (x, y, z)
EPS = 1e-3


def make_state_dict(seed=0, in_channels=5):
    """Random weights under the reference's parameter names (spconv weight [kx,ky,kz,Cin,Cout])."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def spw(name, k, ci, co):
        fan = ci * int(np.prod(k))
        sd[name] = (torch.rand(*k, ci, co, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5

    def bn(prefix, c):
        sd[prefix + ".weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
        sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[prefix + ".running_var"] = torch.rand(c, generator=g) + 0.5

    e = "encoders.lidar.backbone."
    spw(e + "conv_input.0.weight", (3, 3, 3), in_channels, 16)
    bn(e + "conv_input.1", 16)
    cin = 16
    for s, chans in enumerate(ENCODER_CHANNELS):
        for j, co in enumerate(chans):
            p = f"{e}encoder_layers.encoder_layer{s + 1}.{j}"
            if j == len(chans) - 1 and s != len(ENCODER_CHANNELS) - 1:
                spw(p + ".0.weight", (3, 3, 3), cin, co)
                bn(p + ".1", co)
            else:
                spw(p + ".conv1.weight", (3, 3, 3), co, co)
                bn(p + ".bn1", co)
                spw(p + ".conv2.weight", (3, 3, 3), co, co)
                bn(p + ".bn2", co)
            cin = co
    spw(e + "conv_out.0.weight", (1, 1, 3), 128, 128)
    bn(e + "conv_out.1", 128)

    def conv2(name, co, ci, k):
        sd[name] = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5

    d = "decoder.backbone.blocks."
    cin = 256
    for b, (co, n) in enumerate(((128, 5), (256, 5))):
        for i in range(n + 1):
            conv2(f"{d}{b}.{3 * i}.weight", co, cin if i == 0 else co, 3)
            bn(f"{d}{b}.{3 * i + 1}", co)
        cin = co
    conv2("decoder.neck.deblocks.0.0.weight", 256, 128, 1)
    bn("decoder.neck.deblocks.0.1", 256)
    sd["decoder.neck.deblocks.1.0.weight"] = torch.randn(256, 256, 2, 2, generator=g) * (2.0 / 1024) ** 0.5
    bn("decoder.neck.deblocks.1.1", 256)
    return sd


def _bn(sd, p, y):
    sh = (1, -1) + (1,) * (y.dim() - 2)
    return (y - sd[p + ".running_mean"].view(sh)) / torch.sqrt(sd[p + ".running_var"].view(sh) + EPS) \
        * sd[p + ".weight"].view(sh) + sd[p + ".bias"].view(sh)


def _conv3(sd, name, x, stride=1, padding=0):
    w = sd[name].permute(4, 3, 0, 1, 2).contiguous()          # [kx,ky,kz,Ci,Co] -> [Co,Ci,kx,ky,kz]
    return F.conv3d(x, w, stride=stride, padding=padding)


def forward(sd, feats, coords, batch, sparse_shape):
    """feats [M,5] f32, coords [M,4] (batch, x, y, z), sparse_shape (X, Y, Z) -> BEV [N,512,X/8,Y/8]."""
    X, Y, Z = [int(v) for v in sparse_shape]
    feats = torch.as_tensor(feats, dtype=torch.float32)
    c = torch.as_tensor(np.asarray(coords), dtype=torch.int64)
    x = torch.zeros(batch, feats.shape[1], X, Y, Z)
    mask = torch.zeros(batch, 1, X, Y, Z)
    x[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    mask[c[:, 0], 0, c[:, 1], c[:, 2], c[:, 3]] = 1.0
    e = "encoders.lidar.backbone."
    x = torch.relu(_bn(sd, e + "conv_input.1", _conv3(sd, e + "conv_input.0.weight", x, 1, 1))) * mask
    for s, chans in enumerate(ENCODER_CHANNELS):
        for j in range(len(chans)):
            p = f"{e}encoder_layers.encoder_layer{s + 1}.{j}"
            if j == len(chans) - 1 and s != len(ENCODER_CHANNELS) - 1:
                pad = DOWN_PADDING[s]
                pad = (pad,) * 3 if isinstance(pad, int) else tuple(pad)
                mask = (F.max_pool3d(mask, 3, 2, pad) > 0).float()
                x = torch.relu(_bn(sd, p + ".1", _conv3(sd, p + ".0.weight", x, 2, pad))) * mask
            else:
                idt = x
                y = torch.relu(_bn(sd, p + ".bn1", _conv3(sd, p + ".conv1.weight", x, 1, 1))) * mask
                y = _bn(sd, p + ".bn2", _conv3(sd, p + ".conv2.weight", y, 1, 1)) * mask
                x = torch.relu(y + idt) * mask
    mask = (F.max_pool3d(mask, (1, 1, 3), (1, 1, 2), 0) > 0).float()
    x = torch.relu(_bn(sd, e + "conv_out.1", _conv3(sd, e + "conv_out.0.weight", x, (1, 1, 2), 0))) * mask
    N, C, H, W, D = x.shape                                   # sparse_encoder.py:126-130
    x = x.permute(0, 1, 4, 2, 3).contiguous().view(N, C * D, H, W)
    outs = []
    d = "decoder.backbone.blocks."
    for b, (stride, n) in enumerate(((1, 5), (2, 5))):
        for i in range(n + 1):
            x = torch.relu(_bn(sd, f"{d}{b}.{3 * i + 1}",
                               F.conv2d(x, sd[f"{d}{b}.{3 * i}.weight"], stride=stride if i == 0 else 1, padding=1)))
        outs.append(x)
    u0 = torch.relu(_bn(sd, "decoder.neck.deblocks.0.1", F.conv2d(outs[0], sd["decoder.neck.deblocks.0.0.weight"])))
    u1 = torch.relu(_bn(sd, "decoder.neck.deblocks.1.1",
                        F.conv_transpose2d(outs[1], sd["decoder.neck.deblocks.1.0.weight"], stride=2)))
    return torch.cat([u0, u1], dim=1)
