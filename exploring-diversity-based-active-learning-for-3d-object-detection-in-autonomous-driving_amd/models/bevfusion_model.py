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
from . import builder
from .bevfusion_camera import ConvFuser, DepthLSSTransform, GeneralizedLSSFPN
from .registry import DETECTORS
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
        # upsample_cfg of the swint configs (configs/nuscenes/det/transfusion/secfpn/camera+lidar/default.yaml:16-18)
        self.camera_neck = GeneralizedLSSFPN([192, 384, 768], 256, 3, upsample_cfg=dict(mode="bilinear", align_corners=False))
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
        -> (embedding [B,512], decoder map [B,180,180,512], decoded boxes per sample or None)."""
        emb, dec, preds = self._run(example, img, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix,
                                    lidar_aug_matrix, timed=timed)
        return emb, dec, (None if preds is None else self.head.get_bboxes(preds))

    def _run(self, example, img, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix,
             book=None, timed=False):
        """-> (embedding, decoder map, raw head predictions or None)."""
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
                lidar_bev, _ = self.lidar.sparse_stage(example, book=book)
                lidar_bev.record_stream(main)
        feats = self.camera_backbone(img.reshape(B * N, *img.shape[2:]))
        mark("camera backbone (Swin-T)")
        fpn = self.camera_neck(list(feats))[0]
        mark("camera neck (LSS-FPN)")
        cam = self.vtransform(fpn.view(B, N, *fpn.shape[1:]), points, lidar2image, cam_intrinsic, camera2lidar,
                              img_aug_matrix, lidar_aug_matrix, calib_key=example.get("calib_key") if isinstance(example, dict) else None)
        # cam is [x, y]; this build's maps are [H = y, W = x]: the fuser's concatenation kernel transposes it on the way
        mark("view transform (depth LSS)")
        if lidar_bev is None:
            lidar_bev, _ = self.lidar.sparse_stage(example, book=book)
        else:
            torch.cuda.current_stream(img.device).wait_stream(self._side_stream(img.device))
        mark("lidar encoder")
        fused = self.fuser([cam, lidar_bev], first_hw_swapped=True)
        mark("fuser")
        dec = self.lidar.neck(fused)
        emb = getattr(self.lidar.neck, "embedding", None)
        if emb is None:
            emb = D.gap_nhwc(dec)
        mark("decoder + embedding")
        preds = None
        if self.head is not None:
            preds = self.head(dec)
            if timed:
                self.head.get_bboxes(preds)            # the decode is part of the stage's time
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


CAMERA_KEYS = ("img", "points", "lidar2image", "camera_intrinsics", "camera2lidar", "img_aug_matrix", "lidar_aug_matrix")


@DETECTORS.register_module
class BEVFusion(BEVFusionCameraLidar):
    """The camera+lidar model behind the det3d detector contract, so that the selectors and ``tools/active_select.py`` sweep
    it like any other detector (BASELINE configs[4]; reference flow ``fusion_models/bevfusion.py:207-305`` ->
    ``heads/bbox/transfusion.py:714-851``; the reference itself has no selector hook under ``bevfusion/``).

    ``BEVFusion(lidar=<FPNVoxelNet cfg without head>, bbox_head=<TransFusionHead cfg or None>, image_size=..., ...)``;
    ``detector(example, return_loss=False, estimate=True)`` -> ``(list[dict(box3d_lidar, scores, label_preds, metadata)],
    middle)`` with ``middle[-1]`` the decoder map (its ``mean(-1).mean(-1)`` is the fused-BEV embedding).  ``example`` is
    the lidar batch of ``DeviceSweepLoader`` plus the camera side (``CAMERA_KEYS``: channels-last images ``[B,N,H,W,3]``,
    per-sample point clouds, the 4 x 4 calibration / augmentation matrices; ``al3d.datasets.CameraLidarSweepLoader``)."""

    def __init__(self, lidar, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, **kwargs):
        lidar_det = builder.build_detector(lidar, train_cfg=train_cfg, test_cfg=test_cfg) if isinstance(lidar, dict) else lidar
        head = builder.build_head(bbox_head) if isinstance(bbox_head, dict) else bbox_head
        super().__init__(lidar_det, head=head, **kwargs)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @property
    def bbox_head(self):
        return self.head

    def prepare(self, example):
        """Index work of the lidar half (sparse-conv rulebook), run one batch ahead by the sweep."""
        return self.lidar.prepare(example)

    def forward(self, example, return_loss=True, finetune=False, book=None, **kwargs):
        if return_loss:
            raise NotImplementedError("al3d implements the inference sweep, not training")
        missing = [k for k in CAMERA_KEYS if k not in example]
        if missing:
            raise KeyError(f"BEVFusion: the example lacks the camera side {missing} (use CameraLidarSweepLoader)")
        from .detectors import NHWCFeature
        emb, dec, preds = self._run(example, example["img"], example["points"], example["lidar2image"],
                                    example["camera_intrinsics"], example["camera2lidar"], example["img_aug_matrix"],
                                    example["lidar_aug_matrix"], book=book)
        metas = example.get("metadata", None) or [None] * dec.shape[0]
        if self.head is None:
            if not kwargs.get("estimate", False):
                raise RuntimeError("this detector was built without a bbox_head: only the estimate=True embedding sweep "
                                   "is available")
            out = [dict(metadata=m) for m in metas]
        else:
            out = self.head.predict(example, preds, self.test_cfg)
        if kwargs.get("estimate", False):
            return out, [NHWCFeature(dec, emb)]
        return out
