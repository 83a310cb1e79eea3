"""CBGS multi-group anchor head (reference det3d/models/bbox_heads/mg_head.py:199-231,386-533,
697-1085).  Parameter names match (``tasks.<t>.conv_{box,cls}.{weight,bias}``); the twelve 1x1
convs run as ONE fused GEMM on the matrix cores and decode/score/top-k/rotated-NMS run in one
device kernel per batch (no host round trips per task and sample)."""
import ctypes
import logging

import torch
from torch import nn

from .. import detector_ops as D
from .. import lib
from ..selector_ops import _ptr, _stream
from .registry import HEADS


class LazyDetections:
    """``list[dict(box3d_lidar, scores, label_preds, metadata)]`` whose device->host hand-off
    (one copy of the per-(sample, task) counts) happens on first access."""

    def __init__(self, boxes=None, scores=None, labels=None, counts=None, done=None, meta=None, launch=None, batch=0):
        # ``launch``: the decode + NMS launch itself, not issued yet (the sweep releases it beside the next batch's dense
        # neck: MultiGroupHead.flush_deferred); anything that reads the detections issues it first
        self._launch = launch
        self._batch = batch
        self._raw_ = None if launch is not None else (boxes, scores, labels, counts, done, meta)
        self._list = None

    def launch(self, gate=None):
        if self._raw_ is None:
            self._raw_ = self._launch(gate)
            self._launch = None

    @property
    def _raw(self):
        self.launch()
        return self._raw_

    def _materialize(self):
        if self._list is None:
            boxes, scores, labels, counts, done, meta = self._raw
            done.synchronize()
            torch.cuda.current_stream(boxes.device).wait_event(done)
            cnt = counts.cpu()
            B, nt = cnt.shape
            out = []
            for b in range(B):
                bb, ss, ll = [], [], []
                for t in range(nt):
                    c = int(cnt[b, t])
                    bb.append(boxes[b, t, :c])
                    ss.append(scores[b, t, :c])
                    ll.append(labels[b, t, :c].long())
                out.append({"box3d_lidar": torch.cat(bb), "scores": torch.cat(ss),
                            "label_preds": torch.cat(ll), "metadata": meta[b]})
            self._list = out
        return self._list

    def frame_entropy(self):
        """[B] mean binary entropy of each frame's kept scores (device tensor; empty frame ->
        NaN), computed from the raw NMS buffers without materialising the detection dicts."""
        from .. import selector_ops as ops
        boxes, scores, labels, counts, done, meta = self._raw
        torch.cuda.current_stream(boxes.device).wait_event(done)
        return ops.frame_entropy(scores, counts)

    def frame_weighted_entropy(self, class_weight):
        """[B] sum over kept boxes of entropy * class_weight[label] (PPAL, ppal_selector.py:99-109)."""
        from .. import selector_ops as ops
        boxes, scores, labels, counts, done, meta = self._raw
        torch.cuda.current_stream(boxes.device).wait_event(done)
        return ops.frame_weighted_entropy(scores, labels, counts, class_weight)

    def __len__(self):
        return self._batch if self._raw_ is None else self._raw_[3].shape[0]

    def __getitem__(self, i):
        return self._materialize()[i]

    def __iter__(self):
        return iter(self._materialize())


@HEADS.register_module
class Head(nn.Module):
    def __init__(self, num_input, num_pred, num_cls, use_dir=False, num_dir=0, header=True, name="",
                 focal_loss_init=False, **kwargs):
        super().__init__()
        if use_dir:
            raise NotImplementedError("direction classifier is not on the sweep path")
        self.use_dir = use_dir
        self.conv_box = nn.Conv2d(num_input, num_pred, 1)
        self.conv_cls = nn.Conv2d(num_input, num_cls, 1)


@HEADS.register_module
class MultiGroupHead(nn.Module):
    def __init__(self, mode="3d", in_channels=[128], norm_cfg=None, tasks=[], weights=[],
                 num_classes=[1], box_coder=None, with_cls=True, with_reg=True,
                 reg_class_agnostic=False, encode_background_as_zeros=True, loss_norm=None,
                 loss_cls=None, use_sigmoid_score=True, loss_bbox=None, encode_rad_error_by_sin=True,
                 loss_aux=None, direction_offset=0.0, name="rpn", logger=None):
        super().__init__()
        assert with_cls or with_reg
        if loss_aux is not None or mode == "bev" or not encode_background_as_zeros:
            raise NotImplementedError("only the configuration used by examples/active/cbgs_*.py is built")
        assert use_sigmoid_score is True
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.num_anchor_per_locs = [2 * n for n in num_classes]
        self.box_coder = box_coder
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.encode_background_as_zeros = encode_background_as_zeros
        self.use_sigmoid_score = use_sigmoid_score
        self.use_direction_classifier = False
        self.box_n_dim = box_coder.code_size
        self.anchor_dim = box_coder.n_dim
        assert self.box_n_dim == 10 and self.anchor_dim == 9 and box_coder.vec_encode and \
            not box_coder.linear_dim, "the device decoder implements the 9-dim vector-angle code"
        self.logger = logger or logging.getLogger("MultiGroupHead")
        self.tasks = nn.ModuleList()
        for nc, na in zip(num_classes, self.num_anchor_per_locs):
            self.tasks.append(Head(in_channels, na * self.box_n_dim, na * nc, header=False))
        self.logger.info("Finish MultiGroupHead Initialization")

    def init_weights(self, pretrained=None):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------ fused 1x1 convs
    def _prepare(self, device):
        if getattr(self, "_packed_dev", None) == (device, D.MATH, D.DENSE):
            return
        # channel order of the fused output: the box regressions of all tasks, then the class logits of all tasks.
        # The score pre-pass of the decode reads ONLY the class logits: kept together (36 of 236 channels for the six
        # CBGS tasks) they are two 128-byte lines of a pixel's 944-byte record instead of pieces of all eight
        # (head_score_kernel 1.04 -> 0.3 ms per batch of 128).  The offsets travel with the C call.
        ws, bs, self._box_off, self._cls_off = [], [], [], []
        off = 0
        for t in self.tasks:
            self._box_off.append(off)
            ws.append(t.conv_box.weight)
            bs.append(t.conv_box.bias)
            off += t.conv_box.out_channels
        for t in self.tasks:
            self._cls_off.append(off)
            ws.append(t.conv_cls.weight)
            bs.append(t.conv_cls.bias)
            off += t.conv_cls.out_channels
        self._ch = off
        self._w, self._wscale = D.pack_dense(
            D.pack_conv_weight(torch.cat([w.detach() for w in ws], dim=0)).to(device), None, 1, 1, 0)
        self._b = torch.cat([b.detach() for b in bs]).float().contiguous().to(device)
        self._packed_dev = (device, D.MATH, D.DENSE)

    def accepts_pair(self, device):
        """True when the fused head convolution runs on the LDS-DMA kernel and can read pair pixels."""
        self._prepare(device)
        import os
        # AL3D_HEAD_PAIR=0 (A/B): the head reads f32 pixels although the neck could hand it pair pixels
        return D.MATH == "f16x3" and D.DPIX == "pair" and getattr(self._w, "kind", None) == "dma" and \
            os.environ.get("AL3D_HEAD_PAIR", "1") != "0"

    def forward(self, x, finetune=False, in_pair=False):
        """x NHWC [B,H,W,512] -> list of per-task dicts with NHWC views
        (``box_preds [B,H,W,na*10]``, ``cls_preds [B,H,W,na*nc]``) like Head.forward
        (mg_head.py:222-231), plus the fused buffer under ``_fused``."""
        self._prepare(x.device)
        fused = D.conv2d_nhwc(x, self._w, self._wscale, self._b, 1, 1, 0, False, io=D.IO_IN_PAIR if in_pair else 0)
        rets = []
        for t, task in enumerate(self.tasks):
            b0, c0 = self._box_off[t], self._cls_off[t]
            rets.append({"box_preds": fused[..., b0:b0 + task.conv_box.out_channels],
                         "cls_preds": fused[..., c0:c0 + task.conv_cls.out_channels],
                         "_fused": fused})
        return rets

    # ------------------------------------------------------------ decode + NMS
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """Returns list[dict(box3d_lidar [K,9], scores [K], label_preds [K] i64, metadata)]
        per sample (mg_head.py:697-803)."""
        fused = preds_dicts[0]["_fused"]
        B, H, W, CH = fused.shape
        dev = fused.device
        nt = len(self.tasks)
        anchors = example["anchors"]
        a_dev = []
        for t in range(nt):
            a = anchors[t]
            a = a[0] if a.dim() == 3 else a          # identical for every sample of the batch
            a_dev.append(a.reshape(-1, self.anchor_dim).float().contiguous().to(dev))
        nms = test_cfg["nms"] if isinstance(test_cfg, dict) else test_cfg.nms
        if not nms["use_rotate_nms"] or nms["use_multi_class_nms"]:
            raise NotImplementedError("only rotate NMS without multi-class NMS is built")
        post = int(nms["nms_post_max_size"])
        IntA = ctypes.c_int * nt
        label_off, acc = [], 0
        for nc in self.num_classes:
            label_off.append(acc)
            acc += nc
        rng = test_cfg["post_center_limit_range"]
        # The decode+NMS kernel is latency-bound (one workgroup per (sample, task)); it runs on a
        # side stream so the next batch's voxelizer / sparse encoder overlap it, and the host
        # only waits for it when somebody actually reads the detections.
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None or self._side.device != dev:
            from ..sweep import masked_stream
            import os
            # AL3D_NMS_CUS=n: decode + NMS (192 one-workgroup problems per batch) on n compute units
            self._side = masked_stream(dev, int(os.environ.get("AL3D_NMS_CUS", "0")),
                                       int(os.environ.get("AL3D_NMS_CU0", "0")))
        ready = torch.cuda.Event()
        ready.record(main)
        meta = example.get("metadata") or [None] * B

        def launch(gate=None):
          with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            if gate is not None:
                self._side.wait_event(gate)
            boxes = torch.empty((B, nt, post, 9), dtype=torch.float32, device=dev)
            scores = torch.empty((B, nt, post), dtype=torch.float32, device=dev)
            labels = torch.empty((B, nt, post), dtype=torch.int32, device=dev)
            counts = torch.empty((B, nt), dtype=torch.int32, device=dev)
            fused.record_stream(self._side)
            for a in a_dev:
                a.record_stream(self._side)
            task_a = IntA(*[a.shape[0] for a in a_dev])
            ws = torch.empty(lib.load().al3d_head_decode_nms_workspace_bytes(B, nt, task_a), dtype=torch.uint8,
                             device=dev)
            lib.call("al3d_head_decode_nms", _ptr(fused), B, H * W, CH, nt,
                     (ctypes.c_void_p * nt)(*[a.data_ptr() for a in a_dev]),
                     task_a, IntA(*self.num_anchor_per_locs),
                     IntA(*self.num_classes), IntA(*self._box_off), IntA(*self._cls_off), IntA(*label_off),
                     float(test_cfg["score_threshold"]), float(nms["nms_iou_threshold"]),
                     int(nms["nms_pre_max_size"]), post, (ctypes.c_float * 6)(*[float(v) for v in rng]),
                     _ptr(boxes), _ptr(scores), _ptr(labels), _ptr(counts), _ptr(ws), self._side.cuda_stream)
            done = torch.cuda.Event()
            done.record(self._side)
          return boxes, scores, labels, counts, done, meta

        if getattr(self, "defer_nms", False):
            det = LazyDetections(launch=launch, batch=B)
            self._deferred = getattr(self, "_deferred", []) + [det]
            return det
        return LazyDetections(*launch())

    def flush_deferred(self, gate=None):
        """Issue the decode + NMS launches that ``predict`` held back (``defer_nms``), after ``gate`` (an event) if given."""
        pending, self._deferred = getattr(self, "_deferred", []), []
        for det in pending:
            det.launch(gate)
