"""GPU suite: TransFusionHead inference against an independent NCHW evaluation of the same
computation (the reference's ``forward_single`` / ``get_bboxes``: bevfusion/mmdet3d/models/heads/bbox/transfusion.py:215-333,
:714-851; transformer.py:71-112; transfusion_bbox_coder.py:37-123) on the same seeded parameters, written with other
primitives: torch convolutions, unfolded peak windows, attention as explicit matrix products.  mmcv / mmdet are not importable and no checkpoint exists offline: parity unpinned, like the other
BEVFusion rows.  What differs between the two sides is the three 3x3 convolutions (this build's channels-last f16x3
kernels against torch fp32), the layout handling around them, and the whole query decoder (token GEMMs, the 16-channel
multi-head attention kernel, LayerNorm kernel -- against explicit torch matrix products); the comparison is: dense heatmap to 1e-4 of its scale, the same 200 proposals (a swap is excused only between scores
closer than 1e-6), decoded boxes to 1e-3."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CFG = dict(num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10, num_decoder_layers=1,
           num_heads=8, nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation="relu",
           common_heads=dict(center=[2, 2], height=[1, 2], dim=[3, 2], rot=[2, 2], vel=[2, 2]),
           test_cfg=dict(dataset="nuScenes", grid_size=[512, 512, 1], out_size_factor=8, voxel_size=[0.075, 0.075],
                         pc_range=[-54.0, -54.0], nms_type=None),
           bbox_coder=dict(pc_range=[-54.0, -54.0], post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                           score_threshold=0.0, out_size_factor=8, voxel_size=[0.075, 0.075], code_size=10))


def _seed_(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=g) + 0.5
        elif isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            fan = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (1.5 / fan) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.05
    return mod.eval()


def _mha(q_in, k_in, v_in, attn, heads):
    """Multi-head attention written out (what transformer.py:190-470 / torch's multi_head_attention_forward compute):
    packed in_proj split in q | k | v thirds, per-head scaled dot product, softmax over keys, out_proj.  Inputs [L, B, C]."""
    C = q_in.shape[-1]
    wq, wk, wv = attn.in_proj_weight.split(C, dim=0)
    bq, bk, bv = attn.in_proj_bias.split(C, dim=0)
    q, k, v = q_in @ wq.T + bq, k_in @ wk.T + bk, v_in @ wv.T + bv
    Lq, B, _ = q.shape
    Lk = k.shape[0]
    d = C // heads

    def heads_first(x, L):
        return x.reshape(L, B * heads, d).transpose(0, 1)                 # [B*heads, L, d]
    qh, kh, vh = heads_first(q, Lq) * d ** -0.5, heads_first(k, Lk), heads_first(v, Lk)
    w = torch.softmax(qh @ kh.transpose(1, 2), dim=-1)
    out = (w @ vh).transpose(0, 1).reshape(Lq, B, C)
    return out @ attn.out_proj.weight.T + attn.out_proj.bias


def _posembed(pe, xyz):
    """transformer.py:14-30 on positions [B, P, 2] -> [B, F, P], in plain torch."""
    c1, bn, _, c2 = pe.position_embedding_head
    x = F.conv1d(xyz.transpose(1, 2).contiguous(), c1.weight, c1.bias)
    x = F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    return F.conv1d(x, c2.weight, c2.bias)


def _prediction_heads(ffn, x):
    """transformer.py:496-575 on x [B, C, P], in plain torch."""
    out = {}
    for name in ffn.heads:
        cm, last = getattr(ffn, name)
        y = F.relu(F.batch_norm(F.conv1d(x, cm.conv.weight), cm.bn.running_mean, cm.bn.running_var, cm.bn.weight, cm.bn.bias,
                                False, 0.0, cm.bn.eps))
        out[name] = F.conv1d(y, last.weight, last.bias)
    return out


def _reference_forward(h, inputs):
    """An independent NCHW evaluation of the head's forward pass (the computation of transfusion.py:215-333 and
    transformer.py:71-112), written with other primitives than the module under test: torch convolutions, peaks through
    an unfolded 3x3 window, attention as explicit matrix products."""
    bs, _, H, W = inputs.shape
    P, C = h.num_proposals, h.num_classes
    feat = F.conv2d(inputs, h.shared_conv.weight, h.shared_conv.bias, padding=1)
    hm0 = h.heatmap_head[0]
    mid = F.relu(F.batch_norm(F.conv2d(feat, hm0.conv.weight, None, padding=1), hm0.bn.running_mean, hm0.bn.running_var,
                              hm0.bn.weight, hm0.bn.bias, False, 0.0, hm0.bn.eps))
    dense = F.conv2d(mid, h.heatmap_head[1].weight, h.heatmap_head[1].bias, padding=1)
    score = dense.sigmoid()
    # a cell is a peak of class c when it equals the maximum of its 3x3 window; the one-cell frame never is
    win = F.unfold(score.reshape(bs * C, 1, H, W), kernel_size=3).amax(dim=1).reshape(bs, C, H - 2, W - 2)
    peak = torch.zeros_like(score, dtype=torch.bool)
    peak[:, :, 1:-1, 1:-1] = score[:, :, 1:-1, 1:-1] == win
    peak[:, 8:10] = True                                                   # pedestrian, traffic cone: every cell
    masked = torch.where(peak, score, torch.zeros_like(score)).reshape(bs, C, H * W)
    top = masked.reshape(bs, -1).argsort(dim=-1, descending=True)[:, :P]
    top_class, top_cell = torch.div(top, H * W, rounding_mode="floor"), top % (H * W)
    flat = feat.reshape(bs, feat.shape[1], H * W)
    query = torch.stack([flat[b][:, top_cell[b]] for b in range(bs)])                       # [B, Ch, P]
    query = query + F.conv1d(F.one_hot(top_class, C).permute(0, 2, 1).float(), h.class_encoding.weight, h.class_encoding.bias)
    cells = h.bev_pos.to(inputs.device)[0]                                  # [H*W, 2]: (x + 0.5, y + 0.5), x-major
    assert torch.equal(cells[:, 0], (torch.arange(H * W, device=inputs.device) // W).float() + 0.5)
    qpos = torch.stack([cells[top_cell[b]] for b in range(bs)])            # [B, P, 2]
    dec = h.decoder[0]
    qpe = _posembed(dec.self_posembed, qpos).permute(2, 0, 1)
    kpe = _posembed(dec.cross_posembed, cells[None].expand(bs, -1, -1)).permute(2, 0, 1)
    q, k = query.permute(2, 0, 1), flat.permute(2, 0, 1)
    q = F.layer_norm(q + _mha(q + qpe, q + qpe, q + qpe, dec.self_attn, h.num_heads), (q.shape[-1],), dec.norm1.weight,
                     dec.norm1.bias, dec.norm1.eps)
    q = F.layer_norm(q + _mha(q + qpe, k + kpe, k + kpe, dec.multihead_attn, h.num_heads), (q.shape[-1],), dec.norm2.weight,
                     dec.norm2.bias, dec.norm2.eps)
    ff = F.linear(F.relu(F.linear(q, dec.linear1.weight, dec.linear1.bias)), dec.linear2.weight, dec.linear2.bias)
    q = F.layer_norm(q + ff, (q.shape[-1],), dec.norm3.weight, dec.norm3.bias, dec.norm3.eps)
    res = _prediction_heads(h.prediction_heads[0], q.permute(1, 2, 0))
    res["center"] = res["center"] + qpos.permute(0, 2, 1)
    res["query_heatmap_score"] = torch.stack([masked[b][:, top_cell[b]] for b in range(bs)])
    res["dense_heatmap"] = dense
    return res, top, top_class


def test_transfusion_head_matches_reference_transcription():
    from al3d.models.transfusion_head import TransFusionHead
    head = _seed_(TransFusionHead(**CFG), 11).to(DEV)
    names = set(head.state_dict())
    for key in ("shared_conv.bias", "heatmap_head.0.conv.weight", "heatmap_head.0.bn.running_var", "heatmap_head.1.bias",
                "class_encoding.weight", "decoder.0.self_attn.in_proj_weight", "decoder.0.multihead_attn.out_proj.bias",
                "decoder.0.self_posembed.position_embedding_head.3.weight", "decoder.0.norm3.weight",
                "prediction_heads.0.center.0.conv.weight", "prediction_heads.0.center.0.bn.weight",
                "prediction_heads.0.heatmap.1.bias", "prediction_heads.0.vel.1.weight"):
        assert key in names, key                                    # the reference's module tree
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 512, 64, 64, generator=g).to(DEV)
    with torch.no_grad():
        ref, ref_top, ref_cls = _reference_forward(head, x)
        got = head(x.permute(0, 2, 3, 1).contiguous())[0]
    assert got["dense_heatmap"].shape == ref["dense_heatmap"].shape == (2, 10, 64, 64)
    scale = float(ref["dense_heatmap"].abs().max())
    assert float((got["dense_heatmap"] - ref["dense_heatmap"]).abs().max()) <= 1e-4 * scale + 1e-5
    # the same proposals: every selected proposal's heatmap score agrees, and so do the query classes (a swap is
    # excused only between scores closer than 2e-6)
    for b in range(2):
        got_scores = got["query_heatmap_score"][b].max(0).values              # score of each selected proposal
        ref_scores = ref["query_heatmap_score"][b].max(0).values
        assert float((got_scores - ref_scores).abs().max()) <= 2e-6, "proposal scores differ (beyond a near-tie swap)"
        assert head.query_labels[b].tolist() == ref_cls[b].tolist() or \
            float((got_scores.sort().values - ref_scores.sort().values).abs().max()) <= 2e-6
    for key in ("center", "height", "dim", "rot", "vel", "heatmap"):
        assert got[key].shape == ref[key].shape
        assert float((got[key] - ref[key]).abs().max()) <= 1e-3 * max(1.0, float(ref[key].abs().max())), key
    # decode: get_bboxes on both prediction sets
    boxes = head.get_bboxes([got])
    head_labels = head.query_labels
    head.query_labels = ref_cls
    ref_boxes = head.get_bboxes([ref])
    head.query_labels = head_labels
    for b in range(2):
        assert boxes[b]["bboxes"].shape == ref_boxes[b]["bboxes"].shape and boxes[b]["bboxes"].shape[1] == 9
        assert boxes[b]["labels"].tolist() == ref_boxes[b]["labels"].tolist()
        assert float((boxes[b]["scores"] - ref_boxes[b]["scores"]).abs().max()) <= 1e-4
        assert float((boxes[b]["bboxes"] - ref_boxes[b]["bboxes"]).abs().max()) <= 2e-3
        assert len(boxes[b]["scores"]) > 50


def test_transfusion_circle_nms_variant_runs_and_filters():
    from al3d.models.transfusion_head import TransFusionHead, circle_nms
    import numpy as np
    cfg = dict(CFG, test_cfg=dict(CFG["test_cfg"], nms_type="circle"))
    head = _seed_(TransFusionHead(**cfg), 21).to(DEV)
    x = torch.randn(1, 64, 64, 512, generator=torch.Generator().manual_seed(22)).to(DEV)
    with torch.no_grad():
        preds = head(x)
        kept = head.get_bboxes(preds)
        head.test_cfg["nms_type"] = None
        allb = head.get_bboxes(preds)
    assert len(kept[0]["scores"]) <= len(allb[0]["scores"])
    # known answers of the circle rule: two centres 0.1 apart (dist^2 = 0.01 <= 0.175) -> the lower score goes
    dets = np.array([[0.0, 0.0, 0.9], [0.1, 0.0, 0.8], [3.0, 0.0, 0.7]], np.float32)
    assert circle_nms(dets, 0.175) == [0, 2]
    assert circle_nms(dets, 0.005) == [0, 1, 2]


@pytest.mark.parametrize("B,Pq,Pk", [(2, 200, 200), (1, 200, 32400), (2, 37, 1500)])
def test_mha16_kernel_matches_float64(B, Pq, Pk):
    """``al3d_tok_mha16_f32`` (online softmax over key chunks, matrix-core products with both operands split) against the
    attention core in float64, bounded by torch's own fp32 evaluation; q / k / v as column slices of wider matrices."""
    from al3d import token_ops as T
    heads, C = 8, 128
    g = torch.Generator().manual_seed(Pq + Pk)
    qm = (torch.randn(B * Pq, 3 * C, generator=g) * 1.3).to(DEV)
    km = (torch.randn(B * Pk, 2 * C, generator=g) * 1.3).to(DEV)
    q, k, v = qm[:, C:2 * C], km[:, :C], km[:, C:]

    def core(dt):
        def hf(x, L):
            return x.to(dt).reshape(B, L, heads, 16).permute(0, 2, 1, 3)
        w = torch.softmax((hf(q, Pq) * 0.25) @ hf(k, Pk).transpose(-1, -2), dim=-1)
        return (w @ hf(v, Pk)).permute(0, 2, 1, 3).reshape(B * Pq, C)
    ref, ref32 = core(torch.float64), core(torch.float32)
    got = T.mha16(q, k, v, B, Pq, Pk, heads, 0.25)
    e, e32 = float((got.double() - ref).abs().max()), float((ref32.double() - ref).abs().max())
    assert e <= 3.0 * e32 + 1e-6 and e <= 1e-5, (e, e32)


@pytest.mark.parametrize("case", ["random", "ties", "k1", "waymo"])
def test_proposal_kernel_against_the_reference_transcription(case):
    """al3d_tf_proposals_f32 against transfusion.py:236-275 written with torch ops (sigmoid, max_pool2d on the interior, the
    free classes, top-P of the masked scores): the same (class, cell) set in the same order wherever scores differ, ties
    resolved towards the smaller flat index, masked scores / query positions / query features of the winners."""
    import torch.nn.functional as F
    from al3d import lib
    from al3d.selector_ops import _ptr, _stream
    g = torch.Generator().manual_seed(len(case))
    B, H, W, C, P, hidden = 2, 23, 31, 10, 200, 16
    logits = torch.randn(B, H, W, C, generator=g) * 2.0
    k, free = 3, (8, 9)
    if case == "ties":
        logits = torch.round(logits * 2) / 2                          # many exactly equal scores, plateaus of equal maxima
    elif case == "k1":
        k = 1
    elif case == "waymo":
        C, free = 3, (1, 2)
        logits = logits[..., :3].contiguous()
    tokens = torch.randn(B * H * W, hidden, generator=g)
    bev_pos = torch.rand(H * W, 2, generator=g) * 30
    cols, bias = torch.randn(C, hidden, generator=g), torch.randn(hidden, generator=g)
    dev = lambda t: t.to(DEV).contiguous()
    out = dict(cls=torch.empty((B, P), dtype=torch.int64, device=DEV), cell=torch.empty((B, P), dtype=torch.int64, device=DEV),
               score=torch.empty((B, C, P), device=DEV), feat=torch.empty((B * P, hidden), device=DEV),
               pos=torch.empty((B * P, 2), device=DEV))
    ws = torch.empty(int(lib.load().al3d_tf_proposals_workspace_bytes(B, H, W, C)), dtype=torch.uint8, device=DEV)
    dl, dt, dp, dc, db = dev(logits), dev(tokens), dev(bev_pos), dev(cols), dev(bias)
    lib.call("al3d_tf_proposals_f32", _ptr(dl), B, H, W, C, k, sum(1 << c for c in free), P, _ptr(dt), hidden, _ptr(dp), _ptr(dc),
             _ptr(db), _ptr(ws), _ptr(out["cls"]), _ptr(out["cell"]), _ptr(out["score"]), _ptr(out["feat"]), _ptr(out["pos"]),
             _stream())
    torch.cuda.synchronize()
    # the transcription (float32 on the device's own sigmoid values would differ in the last bit: compare via the kernel's
    # masked scores where exactness matters, via torch's where a tolerance is stated)
    score = logits.permute(0, 3, 1, 2).sigmoid()
    r = k // 2
    peak = torch.zeros_like(score, dtype=torch.bool)
    if r > 0:
        peak[:, :, r:H - r, r:W - r] = score[:, :, r:H - r, r:W - r] == F.max_pool2d(score, k, 1, 0)
    else:
        peak[:] = True
    for c in free:
        peak[:, c] = True
    masked = (score * peak).reshape(B, C * H * W)
    cls, cell, sc = out["cls"].cpu(), out["cell"].cpu(), out["score"].cpu()
    flat = cls * (H * W) + cell
    for b in range(B):
        got = masked[b][flat[b]]
        # (1) winners carry the P largest masked scores (as a multiset, to the sigmoid's last bits), in descending order
        want = masked[b].sort(descending=True).values[:P]
        assert torch.allclose(got, want, rtol=0, atol=2e-7), case
        assert bool((got[:-1] >= got[1:] - 2e-7).all())
        assert len(set(flat[b].tolist())) == P
        # (2) equal scores: ascending flat index (checked on the kernel's own score values)
        own = sc[b].reshape(C, P)[cls[b], torch.arange(P)]
        eq = own[:-1] == own[1:]
        assert bool((flat[b][:-1][eq] < flat[b][1:][eq]).all())
        # (3) every proposal is a masked-in cell unless the pool of peaks ran out
        assert bool(((got > 0) | (want <= 0)).all())
        # (4) the C masked scores of each winning cell
        ref_sc = masked[b].reshape(C, H * W)[:, cell[b]]
        assert torch.allclose(sc[b], ref_sc, rtol=0, atol=2e-7)
    # (5) query positions and features of the winners (exact: gathers and two float32 additions)
    pos, feat = out["pos"].cpu().view(B, P, 2), out["feat"].cpu().view(B, P, hidden)
    for b in range(B):
        assert torch.equal(pos[b], bev_pos[cell[b]])
        assert torch.equal(feat[b], tokens[b * H * W + cell[b]] + (cols[cls[b]] + bias))
