"""Detector graphs (reference det3d/models/detectors/{single_stage,voxelnet}.py).

Call contract kept: ``detector(example, return_loss=False, estimate=True)`` returns
``(list[dict(box3d_lidar, scores, label_preds, metadata)], middle)`` with ``middle[-1]`` the
neck output (voxelnet.py:73-81,115-116).  Activations are channels-last in this build, so
``middle[-1]`` is wrapped in ``NHWCFeature`` whose ``mean(-1).mean(-1)`` -- the expression the
selectors apply to it (feature_selector.py:68-71) -- runs the device GAP kernel.
"""
import torch
from torch import nn

from .. import detector_ops as D
from . import builder
from .registry import DETECTORS


class NHWCFeature:
    """A ``[B,C,H,W]``-shaped view of an NHWC buffer for the selectors' embedding tap."""

    def __init__(self, nhwc, embedding=None, pair=False):
        self._nhwc, self._pair = nhwc, pair   # pair: the buffer holds pair pixels (csrc/sp_rows.h), converted on access
        self._emb = embedding           # [B,C] global average the neck already produced (fused GAP), or None
        B, H, W, C = nhwc.shape
        self.shape = torch.Size((B, C, H, W))
        self.device = nhwc.device
        self._w_reduced = False

    @property
    def nhwc(self):
        if self._pair:
            C = self._nhwc.shape[-1]
            self._nhwc, self._pair = D.rows_convert(self._nhwc.view(-1, C), False).view_as(self._nhwc), False
        return self._nhwc

    def mean(self, dim=-1):
        if not self._w_reduced:
            r = NHWCFeature.__new__(NHWCFeature)
            r._nhwc, r._pair, r.shape, r.device, r._w_reduced, r._emb = \
                self._nhwc, self._pair, self.shape[:3], self.device, True, self._emb
            return r
        if self._emb is not None:
            return self._emb                   # emitted by the neck's deblock launches (al3d_*_gap)
        return D.gap_nhwc(self.nhwc)          # mean over W, then over H, in one kernel

    def nchw(self):
        return self.nhwc.permute(0, 3, 1, 2)


@DETECTORS.register_module
class SingleStageDetector(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        # bbox_head=None: embedding-only graph (BEVFusion lidar branch, models/bevfusion_compat.py)
        self.bbox_head = builder.build_head(bbox_head) if bbox_head is not None else None
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None


@DETECTORS.register_module
class FPNVoxelNet(SingleStageDetector):
    def extract_feat(self, data):
        if data.get("mean_features") is not None:     # device voxelizer already reduced the points
            input_features = data["mean_features"]
        else:
            input_features = self.reader(data["features"], data["num_voxels"])
        x, middle = self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"],
                                  **self._cap_kw(data.get("voxel_cap", 0)))
        if self.with_neck:
            x = self.neck(x)
            middle.append(NHWCFeature(x, getattr(self.neck, "embedding", None)))
        return x, middle

    # The forward pass is exposed in two halves so that the sweep can run the sparse half of batch
    # i+1 (voxel features -> sparse encoder -> dense BEV; latency-bound gathers) on one HIP stream
    # while the dense half of batch i (neck + head + decode; matrix-core bound) runs on another.
    def _cap_kw(self, cap):
        """``voxel_cap`` of a device-voxelized example (frames concatenated in order, at most that many rows each) as the
        sparse encoder's ``frame_rows_max`` promise, for encoders that take it."""
        import inspect
        if cap and "frame_rows_max" in inspect.signature(self.backbone.forward).parameters:
            return {"frame_rows_max": int(cap)}
        return {}

    def prepare(self, example):
        """Index work of a batch (sparse-conv rulebook); needs ``coordinates`` only, so the sweep runs
        it one batch ahead on a side stream.  Pass the result as ``book=`` to forward / sparse_stage."""
        if not hasattr(self.backbone, "rulebook_for"):
            return None
        return self.backbone.rulebook_for(example["coordinates"], len(example["num_voxels"]),
                                          example["shape"][0], **self._cap_kw(example.get("voxel_cap", 0)))

    def sparse_stage(self, example, book=None):
        num_voxels = example["num_voxels"]
        data = dict(features=example.get("voxels"), num_voxels=example.get("num_points"),
                    mean_features=example.get("voxel_features"), coors=example["coordinates"],
                    batch_size=len(num_voxels), input_shape=example["shape"][0], voxel_cap=example.get("voxel_cap", 0))
        if data.get("mean_features") is not None:     # device voxelizer already reduced the points
            input_features = data["mean_features"]
        else:
            input_features = self.reader(data["features"], data["num_voxels"])
        if book is not None:
            return self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"], book=book)
        return self.backbone(input_features, data["coors"], data["batch_size"], data["input_shape"],
                             **self._cap_kw(data["voxel_cap"]))

    def dense_stage(self, example, x, middle, finetune=False, **kwargs):
        pair = False
        if self.with_neck:
            want = self.bbox_head is not None and hasattr(self.bbox_head, "accepts_pair") and \
                self.bbox_head.accepts_pair(x.device)
            import inspect
            want = want and "out_pair" in inspect.signature(self.neck.forward).parameters
            x = self.neck(x, out_pair=True) if want else self.neck(x)
            pair = bool(want and getattr(self.neck, "last_out_pair", False))
            middle.append(NHWCFeature(x, getattr(self.neck, "embedding", None), pair=pair))
        if self.bbox_head is None:
            if not kwargs.get("estimate", False):
                raise RuntimeError("this detector was built without a bbox_head: only the estimate=True "
                                   "embedding sweep is available")
            return [dict(metadata=m) for m in example.get("metadata", [None] * x.shape[0])], middle
        preds = self.bbox_head(x, finetune=finetune, in_pair=True) if pair else self.bbox_head(x, finetune=finetune)
        if kwargs.get("get_preds", False):
            return preds
        out = self.bbox_head.predict(example, preds, self.test_cfg)
        if kwargs.get("estimate", False):
            return out, middle
        return out

    def forward(self, example, return_loss=True, finetune=False, **kwargs):
        if return_loss:
            raise NotImplementedError("al3d implements the inference sweep, not training")
        x, middle = self.sparse_stage(example, book=kwargs.pop("book", None))
        return self.dense_stage(example, x, middle, finetune=finetune, **kwargs)


@DETECTORS.register_module
class VoxelNet(FPNVoxelNet):
    """Same graph; ``forward`` returns detections only (voxelnet.py:8-55)."""

    def forward(self, example, return_loss=True, **kwargs):
        kwargs.pop("estimate", None)
        return super().forward(example, return_loss=return_loss, **kwargs)
