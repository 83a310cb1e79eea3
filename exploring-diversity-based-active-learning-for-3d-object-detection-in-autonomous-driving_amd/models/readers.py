"""Voxel feature reader (reference det3d/models/readers/voxel_encoder.py:198-211)."""
import torch
from torch import nn

from .. import lib
from ..selector_ops import _ptr, _stream
from .registry import READERS


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    """Mean of the points of each voxel.  ``forward(features [M,T,F], num_voxels [M])``
    like the reference; the device voxelizer already emits the mean, in which case the
    detector skips this module."""

    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        if not features.is_cuda:
            raise lib.Al3dError("VoxelFeatureExtractorV3: device tensors required (no CPU fallback)")
        f = features[:, :, : self.num_input_features].contiguous().float()
        num = num_voxels.to(torch.int32).contiguous()
        m, t, c = f.shape
        out = torch.empty((m, c), dtype=torch.float32, device=f.device)
        lib.call("al3d_vfe_mean_f32", _ptr(f), _ptr(num), m, t, c, _ptr(out), _stream())
        return out
