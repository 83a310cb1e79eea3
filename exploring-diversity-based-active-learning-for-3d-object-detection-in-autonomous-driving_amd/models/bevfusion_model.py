"""BEVFusion camera+lidar model assembled from this build's modules (BASELINE configs[4]).

Reference: bevfusion/mmdet3d/models/fusion_models/bevfusion.py:24-305 (``BEVFusion.forward_single``): camera encoder
(backbone -> neck -> vtransform) and lidar encoder (voxelize -> sparse backbone) each produce a BEV map, the fuser
merges them, the decoder (SECOND + SECONDFPN) refines, the head decodes; the embedding the diversity selectors need is
the global average of the decoder neck's output (the reference has no such tap).

The lidar half is an ``FPNVoxelNet``-shaped detector built without a head (``examples/active/
bevfusion_lidar_spatial_temporal_feature.py``): its sparse stage yields the lidar BEV map, its neck IS the decoder.  Maps
here are channels-last with [H=y, W=x]; the camera BEV map comes out of the view transform as [x, y] and is transposed.
Seeded random weights only (no checkpoint offline); parity of every module: see their own tests, all unpinned.
"""
import torch
from torch import nn

from .. import detector_ops as D
from .bevfusion_camera import ConvFuser, DepthLSSTransform, GeneralizedLSSFPN
from .swin import SwinTransformer
from .transfusion_head import TransFusionHead


class BEVFusionCameraLidar(nn.Module):
    def __init__(self, lidar_detector, image_size=(256, 704), feature_size=(32, 88), xbound=(-54.0, 54.0, 0.3),
                 ybound=(-54.0, 54.0, 0.3), zbound=(-10.0, 10.0, 20.0), dbound=(1.0, 60.0, 0.5), camera_channels=80,
                 lidar_channels=256, head=None):
        super().__init__()
        self.lidar = lidar_detector                                   # sparse encoder + SECOND / SECONDFPN ("neck")
        self.camera_backbone = SwinTransformer(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7,
                                               mlp_ratio=4, qkv_bias=True, patch_norm=True, out_indices=[1, 2, 3])
        self.camera_neck = GeneralizedLSSFPN([192, 384, 768], 256, 3)
        self.vtransform = DepthLSSTransform(256, camera_channels, image_size, feature_size, list(xbound), list(ybound),
                                            list(zbound), list(dbound), downsample=2)
        self.fuser = ConvFuser([camera_channels, lidar_channels], lidar_channels)
        self.head = head
        self.stage_ms = None
        import os
        self.overlap = os.environ.get("AL3D_BEV_OVERLAP", "0") == "1"     # lidar encoder on a side stream beside the camera branch (measured neutral: 205-209 frames/s either way)
        object.__setattr__(self, "_side", {})

    def _side_stream(self, device):
        key = torch.device(device).index or 0
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def forward(self, example, img, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix,
                timed=False):
        """example: the lidar batch (``DeviceSweepLoader`` dict); img [B,N,H,W,3] channels-last.
        -> (embedding [B,512], decoder map [B,180,180,512], head predictions or None)."""
        B, N = img.shape[:2]
        marks = []

        def mark(name):
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))
        mark("start")
        # The two encoders meet only at the fuser: outside the per-stage timing mode the lidar encoder (gather-bound sparse
        # kernels) runs on a second stream beside the camera branch (matrix-core / HBM-bound token and conv kernels).
        lidar_bev = None
        if not timed and img.is_cuda and self.overlap:
            main = torch.cuda.current_stream(img.device)
            side = self._side_stream(img.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lidar_bev, _ = self.lidar.sparse_stage(example)
                lidar_bev.record_stream(main)
        feats = self.camera_backbone(img.reshape(B * N, *img.shape[2:]))
        mark("camera backbone (Swin-T)")
        fpn = self.camera_neck(list(feats))[0]
        mark("camera neck (LSS-FPN)")
        cam = self.vtransform(fpn.view(B, N, *fpn.shape[1:]), points, lidar2image, cam_intrinsic, camera2lidar,
                              img_aug_matrix, lidar_aug_matrix)
        cam = cam.permute(0, 2, 1, 3).contiguous()                   # [x, y] -> this build's [H=y, W=x]
        mark("view transform (depth LSS)")
        if lidar_bev is None:
            lidar_bev, _ = self.lidar.sparse_stage(example)
        else:
            torch.cuda.current_stream(img.device).wait_stream(self._side_stream(img.device))
        mark("lidar encoder")
        fused = self.fuser([cam, lidar_bev])
        mark("fuser")
        dec = self.lidar.neck(fused)
        emb = getattr(self.lidar.neck, "embedding", None)
        if emb is None:
            emb = D.gap_nhwc(dec)
        mark("decoder + embedding")
        preds = None
        if self.head is not None:
            out = self.head(dec)
            preds = self.head.get_bboxes(out)
            mark("TransFusionHead")
        if timed:
            torch.cuda.synchronize()
            self.stage_ms = {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(marks[:-1], marks[1:])}
        return emb, dec, preds


def transfusion_head_for(grid=1440, in_channels=512):
    """The reference's head configuration (configs/nuscenes/det/transfusion/default.yaml) for a grid x grid BEV."""
    return TransFusionHead(
        num_proposals=200, auxiliary=True, in_channels=in_channels, hidden_channel=128, num_classes=10, num_decoder_layers=1,
        num_heads=8, nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation="relu",
        transpose_input=True,          # the decoder map here is [H = y, W = x]; the head works on the reference's [x, y]
        common_heads=dict(center=[2, 2], height=[1, 2], dim=[3, 2], rot=[2, 2], vel=[2, 2]),
        test_cfg=dict(dataset="nuScenes", grid_size=[grid, grid, 1], out_size_factor=8, voxel_size=[0.075, 0.075],
                      pc_range=[-54.0, -54.0], nms_type=None),
        bbox_coder=dict(pc_range=[-54.0, -54.0], post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        score_threshold=0.0, out_size_factor=8, voxel_size=[0.075, 0.075], code_size=10))
