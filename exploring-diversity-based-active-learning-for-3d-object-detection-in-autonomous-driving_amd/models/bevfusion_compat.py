"""BEVFusion lidar branch on the det3d-shaped model of this build (BASELINE configs[3]).

BEVFusion's lidar encoder + decoder (voxelnet_0p075: hard voxelisation + mean, ``SparseEncoder`` with
basic blocks, ``SECOND`` [5,5] layers, ``SECONDFPN`` 2 x 256) is, layer for layer, the graph of the
reference's CBGS ``FPNVoxelNet`` (``FPNSpMiddleResNetFHD`` + ``RPN``) on a 0.075 m / 1440 x 1440 x 41
grid: same channel plan, same strided-conv paddings ((1,1,0) in their (x,y,z) order = (0,1,1) in
(z,y,x)), same (1,1,3)/(1,1,2) output conv over z, same C-major ``[C*D,H,W]`` flattening
(bevfusion/mmdet3d/models/backbones/sparse_encoder.py:57-130; configs/nuscenes/det/transfusion/
secfpn/lidar/voxelnet_0p075.yaml, .../secfpn/default.yaml; det3d/models/backbones/scn.py:316-392).
What differs is naming and axis order: their sparse tensors are indexed (batch, x, y, z) and their BEV
maps are [N,C,H=x,W=y]; this build keeps det3d's (batch, z, y, x) / H=y, W=x.  ``convert_lidar_state_dict``
renames a BEVFusion checkpoint's lidar-branch parameters and permutes the kernel axes accordingly, so
the BEV map comes out transposed in (H, W) -- which the selectors' global average does not see.

Only embeddings are produced for this config (``bbox_head=None``): the TransFusion head is not part of
the hot path of any selector in the reference (nothing under bevfusion/ calls a selector).
"""
import torch

_STAGE_SLOT = {                      # (encoder_layer index s, j) -> (middle_conv index, position)
    (1, 0): (0, 3), (1, 1): (0, 4), (1, 2): (0, 5),
    (2, 0): (1, 0), (2, 1): (1, 1), (2, 2): (1, 2),
    (3, 0): (2, 0), (3, 1): (2, 1), (3, 2): (2, 2),
    (4, 0): (3, 0), (4, 1): (3, 1),
}
_BN = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")


def _sp(w):
    """spconv weight [kx,ky,kz,Cin,Cout] -> [kz,ky,kx,Cin,Cout]."""
    return w.permute(2, 1, 0, 3, 4).contiguous()


def _t2(w):
    """2-D kernels act on [H=x,W=y] maps there and on [H=y,W=x] maps here."""
    return w.transpose(2, 3).contiguous()


def convert_lidar_state_dict(sd, lidar_prefix="encoders.lidar.backbone.", decoder_prefix="decoder."):
    """BEVFusion state dict -> state dict of this build's ``FPNVoxelNet(bbox_head=None)``.
    Raises KeyError if a parameter of the lidar branch is missing; camera / fuser / head entries are
    ignored.  The basic blocks' convs have no bias there (mmdet BasicBlock, bias=False) and one
    here (det3d scn.py:68-73): filled with zeros."""
    out = {}

    def bn(src, dst):
        for k in _BN:
            if src + "." + k in sd:
                out[dst + "." + k] = sd[src + "." + k]
            elif k != "num_batches_tracked":
                raise KeyError(src + "." + k)

    e = lidar_prefix
    out["backbone.middle_conv0.0.weight"] = _sp(sd[e + "conv_input.0.weight"])
    bn(e + "conv_input.1", "backbone.middle_conv0.1")
    for (s, j), (mc, pos) in _STAGE_SLOT.items():
        src = f"{e}encoder_layers.encoder_layer{s}.{j}"
        dst = f"backbone.middle_conv{mc}.{pos}"
        if j == 2:                                   # strided SparseConv3d + BN
            out[dst + ".weight"] = _sp(sd[src + ".0.weight"])
            bn(src + ".1", f"backbone.middle_conv{mc}.{pos + 1}")
        else:                                        # SparseBasicBlock
            for c in ("1", "2"):
                w = _sp(sd[f"{src}.conv{c}.weight"])
                out[f"{dst}.conv{c}.weight"] = w
                out[f"{dst}.conv{c}.bias"] = torch.zeros(w.shape[-1], dtype=w.dtype)
                bn(f"{src}.bn{c}", f"{dst}.bn{c}")
    out["backbone.middle_conv3.2.weight"] = _sp(sd[e + "conv_out.0.weight"])
    bn(e + "conv_out.1", "backbone.middle_conv3.3")
    out.update(convert_decoder_state_dict(sd, decoder_prefix))
    return out


def convert_decoder_state_dict(sd, decoder_prefix="decoder.", neck_prefix="neck."):
    """The decoder half alone: BEVFusion's ``SECOND`` + ``SECONDFPN`` parameters (``<decoder_prefix>backbone.blocks.b.k.*``,
    ``<decoder_prefix>neck.deblocks.b.k.*``; second.py:42-70, necks/second.py:46-75) -> this build's ``RPN``
    (``<neck_prefix>blocks.b.k.*`` with the ZeroPad2d slot, ``<neck_prefix>deblocks.b.k.*``), 2-D kernels transposed for the
    [H = y, W = x] maps.  Pinned by tests/test_bevfusion_second_golden_gpu.py against the reference classes' own output."""
    out = {}

    def bn(src, dst):
        for k in _BN:
            if src + "." + k in sd:
                out[dst + "." + k] = sd[src + "." + k]
            elif k != "num_batches_tracked":
                raise KeyError(src + "." + k)

    d = decoder_prefix
    b = 0
    while f"{d}backbone.blocks.{b}.0.weight" in sd:
        i = 0
        while f"{d}backbone.blocks.{b}.{3 * i}.weight" in sd:     # there: conv, bn, relu; here: pad, conv, bn, relu, ...
            out[f"{neck_prefix}blocks.{b}.{3 * i + 1}.weight"] = _t2(sd[f"{d}backbone.blocks.{b}.{3 * i}.weight"])
            bn(f"{d}backbone.blocks.{b}.{3 * i + 1}", f"{neck_prefix}blocks.{b}.{3 * i + 2}")
            i += 1
        out[f"{neck_prefix}deblocks.{b}.0.weight"] = _t2(sd[f"{d}neck.deblocks.{b}.0.weight"])
        bn(f"{d}neck.deblocks.{b}.1", f"{neck_prefix}deblocks.{b}.1")
        b += 1
    if b == 0:
        raise KeyError(f"{d}backbone.blocks.0.0.weight")
    return out


def to_bevfusion_coords(coords_bzyx):
    """(batch, z, y, x) -> (batch, x, y, z)."""
    return coords_bzyx[:, [0, 3, 2, 1]]
