"""TransFusionHead, inference half (BASELINE configs[3]-[4]).

Reference: bevfusion/mmdet3d/models/heads/bbox/transfusion.py:34-380 (``forward_single``), :714-851 (``get_bboxes``),
models/utils/transformer.py:14-112,496-575 (``PositionEmbeddingLearned``, ``TransformerDecoderLayer``, ``FFN``),
core/bbox/coders/transfusion_bbox_coder.py:37-123 (``decode``), core/post_processing/box3d_nms.py:181-222
(``circle_nms``).  Training (targets, Hungarian assignment, losses) is out of scope: the sweep runs ``eval()`` only.

What runs where: the three 3x3 convolutions over the BEV map (shared conv 512 -> 128, heatmap head 128 -> 128 -> 10:
~96 % of the head's flops at 180 x 180) go through this build's dense conv kernels on channels-last maps; the
query side -- 200 proposals x 128 channels: class encoding, position embeddings, self / cross attention, FFN and
prediction heads -- is a few small library GEMMs and stays on torch ops, as does the top-k over the heatmap.
Parameter names follow the reference module tree, so its state dicts load.  mmcv's ``ConvModule`` gives
``<name>.conv`` / ``<name>.bn``; ``bias="auto"`` handed to a bare conv layer is truthy, i.e. a bias (the reference's
``shared_conv`` and ``heatmap_head.1`` have one).
"""
import copy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .bevfusion_camera import _ConvAffine
from .registry import HEADS


class PositionEmbeddingLearned(nn.Module):
    """transformer.py:14-30."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1), nn.BatchNorm1d(num_pos_feats), nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class TransformerDecoderLayer(nn.Module):
    """transformer.py:33-112 (``cross_only=False``); ``nn.MultiheadAttention`` is the module the reference copied
    (same parameters: in_proj_weight / in_proj_bias / out_proj)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", self_posembed=None,
                 cross_posembed=None):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.activation = {"relu": F.relu, "gelu": F.gelu}[activation]
        self.self_posembed, self.cross_posembed = self_posembed, cross_posembed

    def forward(self, query, key, query_pos, key_pos):
        """query [B,C,Pq], key [B,C,Pk], query_pos [B,Pq,2], key_pos [B,Pk,2] -> [B,C,Pq] (eval: dropouts are identity)."""
        qpe = self.self_posembed(query_pos).permute(2, 0, 1)
        kpe = self.cross_posembed(key_pos).permute(2, 0, 1)
        query, key = query.permute(2, 0, 1), key.permute(2, 0, 1)
        q = k = v = query + qpe
        query = self.norm1(query + self.self_attn(q, k, value=v)[0])
        kk = key + kpe
        query = self.norm2(query + self.multihead_attn(query=query + qpe, key=kk, value=kk)[0])
        query = self.norm3(query + self.linear2(self.activation(self.linear1(query))))
        return query.permute(1, 2, 0)


class _ConvModule1d(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1, bias=False)
        self.bn = nn.BatchNorm1d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class FFN(nn.Module):
    """transformer.py:496-575: one small Conv1d stack per regression target."""

    def __init__(self, in_channels, heads, head_conv=64, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(_ConvModule1d(c_in, head_conv))
                c_in = head_conv
            layers.append(nn.Conv1d(head_conv, classes, 1, bias=True))
            setattr(self, head, nn.Sequential(*layers))
        getattr(self, "heatmap")[-1].bias.data.fill_(init_bias)

    def forward(self, x):
        return {head: getattr(self, head)(x) for head in self.heads}


class _ConvModule2d(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(cout)


def circle_nms(dets, thresh, post_max_size=83):
    """Greedy centre-distance suppression (the rule of box3d_nms.py:181-222): visit detections [x, y, score] by
    descending score; a visited, unsuppressed one is kept and suppresses every later one whose squared BEV distance to
    it is <= thresh.  Returns the kept indices in visiting order (at most post_max_size)."""
    dets = np.asarray(dets)
    n = dets.shape[0]
    if n == 0:
        return []
    rank = dets[:, 2].argsort()[::-1]                     # ties resolve as in the reference's reversed ascending sort
    xy = dets[rank, :2]
    alive = np.ones(n, dtype=bool)
    kept = []
    for pos in range(n):
        if not alive[pos]:
            continue
        kept.append(int(rank[pos]))
        if pos + 1 < n:
            d2 = (xy[pos + 1:, 0] - xy[pos, 0]) ** 2 + (xy[pos + 1:, 1] - xy[pos, 1]) ** 2
            alive[pos + 1:] &= ~(d2 <= thresh)
    return kept[:post_max_size]


@HEADS.register_module
class TransFusionHead(nn.Module):
    """Inference restatement of the reference head.  ``forward(x)``: x channels-last BEV map [B,H,W,in_channels] ->
    ``[dict]`` with the reference's keys (center, height, dim, rot, vel, heatmap [B,*,P]; query_heatmap_score,
    dense_heatmap [B,num_classes,H,W]); ``get_bboxes(preds)`` -> per sample dict(bboxes [K,9|7], scores, labels)."""

    def __init__(self, num_proposals=128, auxiliary=True, in_channels=128 * 3, hidden_channel=128, num_classes=4,
                 num_decoder_layers=3, num_heads=8, nms_kernel_size=1, ffn_channel=256, dropout=0.1, bn_momentum=0.1,
                 activation="relu", common_heads=None, num_heatmap_convs=2, test_cfg=None, bbox_coder=None, **_unused):
        super().__init__()
        self.num_classes, self.num_proposals, self.auxiliary = num_classes, num_proposals, auxiliary
        self.in_channels, self.num_heads, self.num_decoder_layers = in_channels, num_heads, num_decoder_layers
        self.nms_kernel_size, self.test_cfg, self.bbox_coder = nms_kernel_size, dict(test_cfg or {}), dict(bbox_coder or {})
        self.shared_conv = nn.Conv2d(in_channels, hidden_channel, 3, padding=1, bias=True)
        self.heatmap_head = nn.Sequential(_ConvModule2d(hidden_channel, hidden_channel),
                                          nn.Conv2d(hidden_channel, num_classes, 3, padding=1, bias=True))
        self.class_encoding = nn.Conv1d(num_classes, hidden_channel, 1)
        self.decoder = nn.ModuleList([
            TransformerDecoderLayer(hidden_channel, num_heads, ffn_channel, dropout, activation,
                                    self_posembed=PositionEmbeddingLearned(2, hidden_channel),
                                    cross_posembed=PositionEmbeddingLearned(2, hidden_channel))
            for _ in range(num_decoder_layers)])
        self.prediction_heads = nn.ModuleList()
        for _ in range(num_decoder_layers):
            heads = copy.deepcopy(dict(common_heads or {}))
            heads.update(dict(heatmap=(num_classes, num_heatmap_convs)))
            self.prediction_heads.append(FFN(hidden_channel, heads))
        for m in self.decoder.parameters():
            if m.dim() > 1:
                nn.init.xavier_uniform_(m)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum
        x_size = self.test_cfg["grid_size"][0] // self.test_cfg["out_size_factor"]
        y_size = self.test_cfg["grid_size"][1] // self.test_cfg["out_size_factor"]
        self.bev_pos = self.create_2D_grid(x_size, y_size)
        object.__setattr__(self, "_convs", [_ConvAffine(self.shared_conv, None, False),
                                            _ConvAffine(self.heatmap_head[0].conv, self.heatmap_head[0].bn, True),
                                            _ConvAffine(self.heatmap_head[1], None, False)])
        self.query_labels = None

    @staticmethod
    def create_2D_grid(x_size, y_size):
        """transfusion.py:173-184."""
        bx, by = torch.meshgrid(torch.linspace(0, x_size - 1, x_size), torch.linspace(0, y_size - 1, y_size), indexing="ij")
        coord_base = torch.cat([(bx + 0.5)[None], (by + 0.5)[None]], dim=0)[None]
        return coord_base.view(1, 2, -1).permute(0, 2, 1)

    def forward(self, x):
        if self.training:
            raise RuntimeError("al3d TransFusionHead implements the eval() path only")
        B, H, W, _ = x.shape
        lidar_nhwc = self._convs[0](x)                                        # [B,H,W,hidden]
        dense_nhwc = self._convs[2](self._convs[1](lidar_nhwc))               # [B,H,W,num_classes]
        lidar_feat_flatten = lidar_nhwc.reshape(B, H * W, -1).permute(0, 2, 1)   # [B,C,H*W] (a view: h-major like .view)
        dense_heatmap = dense_nhwc.permute(0, 3, 1, 2)                        # [B,num_classes,H,W]
        bev_pos = self.bev_pos.repeat(B, 1, 1).to(x.device)
        top_class, top_index, heatmap = self._proposals(dense_heatmap)
        query_feat = lidar_feat_flatten.gather(index=top_index[:, None, :].expand(-1, lidar_feat_flatten.shape[1], -1), dim=-1)
        self.query_labels = top_class
        one_hot = F.one_hot(top_class, num_classes=self.num_classes).permute(0, 2, 1)
        query_feat = query_feat + self.class_encoding(one_hot.float())
        query_pos = bev_pos.gather(index=top_index[:, None, :].permute(0, 2, 1).expand(-1, -1, bev_pos.shape[-1]), dim=1)
        ret_dicts = []
        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, lidar_feat_flatten, query_pos, bev_pos)
            res = self.prediction_heads[i](query_feat)
            res["center"] = res["center"] + query_pos.permute(0, 2, 1)
            ret_dicts.append(res)
            query_pos = res["center"].detach().clone().permute(0, 2, 1)
        ret_dicts[0]["query_heatmap_score"] = heatmap.gather(index=top_index[:, None, :].expand(-1, self.num_classes, -1),
                                                             dim=-1)
        ret_dicts[0]["dense_heatmap"] = dense_heatmap
        if self.auxiliary is False:
            return [ret_dicts[-1]]
        new_res = {}
        for key in ret_dicts[0].keys():
            if key not in ("dense_heatmap", "dense_heatmap_old", "query_heatmap_score"):
                new_res[key] = torch.cat([r[key] for r in ret_dicts], dim=-1)
            else:
                new_res[key] = ret_dicts[0][key]
        return [new_res]

    def _proposals(self, dense_heatmap):
        """Query initialisation (transfusion.py:236-275): a cell proposes class c when its sigmoid score is the maximum
        of its k x k neighbourhood -- evaluated on the interior only: the k//2-wide frame never proposes -- except for
        the small-object classes (nuScenes pedestrian / traffic cone, Waymo pedestrian / cyclist), where every cell may;
        the num_proposals best (class, cell) pairs over all classes win.  -> (class [B,P], cell [B,P], masked scores
        [B,C,H*W])."""
        score = dense_heatmap.detach().sigmoid()
        B, C, H, W = score.shape
        k = self.nms_kernel_size
        r = k // 2
        peak = torch.zeros_like(score, dtype=torch.bool)
        if r > 0:
            inner = score[:, :, r:H - r, r:W - r]
            peak[:, :, r:H - r, r:W - r] = inner == F.max_pool2d(score, kernel_size=k, stride=1, padding=0)
        else:
            peak[:] = True
        free = {"nuScenes": (8, 9), "Waymo": (1, 2)}.get(self.test_cfg.get("dataset"), ())
        for c in free:
            peak[:, c] = True
        masked = (score * peak).reshape(B, C, H * W)
        order = masked.reshape(B, -1).argsort(dim=-1, descending=True)[..., :self.num_proposals]
        return order // (H * W), order % (H * W), masked

    # ---------------------------------------------------------------- decode
    def _decode(self, heatmap, rot, dim, center, height, vel):
        """TransFusionBBoxCoder.decode(..., filter=True) (transfusion_bbox_coder.py:37-123)."""
        c = self.bbox_coder
        final_preds = heatmap.max(1).indices
        final_scores = heatmap.max(1).values
        center = center.clone()
        dim = dim.clone()
        center[:, 0, :] = center[:, 0, :] * c["out_size_factor"] * c["voxel_size"][0] + c["pc_range"][0]
        center[:, 1, :] = center[:, 1, :] * c["out_size_factor"] * c["voxel_size"][1] + c["pc_range"][1]
        dim[:, 0, :], dim[:, 1, :], dim[:, 2, :] = dim[:, 0, :].exp(), dim[:, 1, :].exp(), dim[:, 2, :].exp()
        height = height - dim[:, 2:3, :] * 0.5
        rot = torch.atan2(rot[:, 0:1, :], rot[:, 1:2, :])
        parts = [center, height, dim, rot] + ([] if vel is None else [vel])
        boxes = torch.cat(parts, dim=1).permute(0, 2, 1)
        rng = torch.tensor(c["post_center_range"], device=heatmap.device)
        mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
        thr = c.get("score_threshold")
        out = []
        for i in range(heatmap.shape[0]):
            cmask = mask[i]
            if thr:                                                           # 0.0 / None: no score filter (reference)
                cmask = cmask & (final_scores[i] > thr)
            out.append(dict(bboxes=boxes[i, cmask], scores=final_scores[i, cmask], labels=final_preds[i, cmask]))
        return out

    def get_bboxes(self, preds_dicts):
        """transfusion.py:714-851 for one feature level: scores = sigmoid(heatmap) x query heatmap score x one-hot of
        the query's class; decode; optional per-group circle NMS (nuScenes: pedestrians and cones, radius 0.175)."""
        p = preds_dicts[0]
        P = self.num_proposals
        score = p["heatmap"][..., -P:].sigmoid()
        one_hot = F.one_hot(self.query_labels, num_classes=self.num_classes).permute(0, 2, 1)
        score = score * p["query_heatmap_score"] * one_hot
        vel = p["vel"][..., -P:] if "vel" in p else None
        temp = self._decode(score, p["rot"][..., -P:], p["dim"][..., -P:], p["center"][..., -P:], p["height"][..., -P:], vel)
        if self.test_cfg.get("nms_type") is None:
            return temp
        if self.test_cfg["nms_type"] != "circle":
            raise NotImplementedError("only nms_type null / circle are built")
        dataset = self.test_cfg.get("dataset", "nuScenes")
        if dataset == "nuScenes":                      # transfusion.py:751-771
            tasks = [dict(indices=[0, 1, 2, 3, 4, 5, 6, 7], radius=-1), dict(indices=[8], radius=0.175),
                     dict(indices=[9], radius=0.175)]
        elif dataset == "Waymo":                       # transfusion.py:772-779
            tasks = [dict(indices=[0], radius=0.7), dict(indices=[1], radius=0.7), dict(indices=[2], radius=0.7)]
        else:
            raise NotImplementedError(f"circle NMS task table for dataset {dataset!r} (the reference knows nuScenes, Waymo)")
        rets = []
        for t in temp:
            boxes3d, scores, labels = t["bboxes"], t["scores"], t["labels"]
            keep_mask = torch.zeros_like(scores)
            for task in tasks:
                task_mask = torch.zeros_like(scores)
                for cls_idx in task["indices"]:
                    task_mask += labels == cls_idx
                task_mask = task_mask.bool()
                if task["radius"] > 0:
                    dets = torch.cat([boxes3d[task_mask][:, :2], scores[:, None][task_mask]], dim=1)
                    keep = torch.tensor(circle_nms(dets.detach().cpu().numpy(), task["radius"]), dtype=torch.long)
                else:
                    keep = torch.arange(int(task_mask.sum()))
                if keep.shape[0] != 0:
                    keep_mask[torch.where(task_mask != 0)[0][keep.to(scores.device)]] = 1
            keep_mask = keep_mask.bool()
            rets.append(dict(bboxes=boxes3d[keep_mask], scores=scores[keep_mask], labels=labels[keep_mask]))
        return rets
