"""GPU suite: the Swin-T image backbone on the token kernels (al3d/models/swin.py, csrc/tokens.hip; mmdet 2.20.0's
``SwinTransformer`` is not in the reference tree -- parity unpinned, see the module docstring).

What is checked, kernel by kernel, through the C ABI:
  * ``al3d_tok_layernorm_f32`` / ``al3d_tok_linear_f16x3`` / ``al3d_tok_window_attention_f32`` against the torch
    restatement (tests/swin_torch.py) evaluated in float64, at the fp32-class bound of
    ``test_conv2d_f16x3_is_fp32_class`` (error <= 1.5e-6 of sum|a b| for the GEMM) and against the same restatement in
    float32 (the kernel must not be further from the float64 value than ~3x torch's own fp32 evaluation);
  * the restatement itself: mmdet-style relative-position index == the published construction; shifted-window attention
    (roll + window partition + additive region mask) == a DENSE attention over all padded tokens with an independently
    derived "same window and same region" mask, on a map that needs padding;
  * one Swin block, patch merging and the whole Swin-T against the restatement at small and full (6 x 256 x 704) size;
    output levels / strides / channels for the BEVFusion configuration, and the camera branch Swin-T ->
    GeneralizedLSSFPN -> DepthLSSTransform end to end."""
import numpy as np
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_relative_position_index_is_the_published_one():
    from al3d.models.swin import WindowMSA
    m = WindowMSA(96, 3, (7, 7))
    coords = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)      # [2, 49]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += 6
    rel[:, :, 1] += 6
    rel[:, :, 0] *= 13
    assert torch.equal(m.relative_position_index, rel.sum(-1))


def test_restatement_shifted_window_attention_equals_dense_masked_attention():
    """Pins the CHECKER (tests/swin_torch.py): its shifted-window attention equals a dense attention over all padded
    tokens with an independently built window / region mask."""
    import swin_torch as R
    from al3d.models.swin import ShiftWindowMSA
    torch.manual_seed(3)
    C, heads, ws, shift = 96, 3, 7, 3
    attn = ShiftWindowMSA(C, heads, ws, shift).to(DEV).eval()
    torch.nn.init.normal_(attn.w_msa.relative_position_bias_table, std=0.5)
    H, W = 16, 23                                                   # padded to 21 x 28
    x = torch.randn(2, H * W, C, device=DEV)
    with torch.no_grad():
        got = R.shift_window_msa(attn, x, (H, W))
        Hp, Wp = 21, 28
        xp = torch.zeros(2, Hp, Wp, C, device=DEV)
        xp[:, :H, :W] = x.view(2, H, W, C)
        hh, ww = torch.meshgrid(torch.arange(Hp, device=DEV), torch.arange(Wp, device=DEV), indexing="ij")
        hs, wsft = (hh - shift) % Hp, (ww - shift) % Wp              # where a token sits after the cyclic shift
        win = (hs // ws) * (Wp // ws) + (wsft // ws)

        def region(v, n):
            return (v >= n - ws).long() + (v >= n - shift).long()
        reg = region(hs, Hp) * 3 + region(wsft, Wp)
        win, reg = win.flatten(), reg.flatten()
        allowed = (win[:, None] == win[None, :]) & (reg[:, None] == reg[None, :])
        ih, iw = (hs % ws).flatten(), (wsft % ws).flatten()
        idx = (ih[:, None] - ih[None, :] + ws - 1) * (2 * ws - 1) + (iw[:, None] - iw[None, :] + ws - 1)
        same_win = win[:, None] == win[None, :]
        m = attn.w_msa
        qkv = m.qkv(xp.view(2, Hp * Wp, C)).reshape(2, Hp * Wp, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        logits = (q * m.scale) @ k.transpose(-2, -1)                                        # [2, heads, L, L]
        bias = m.relative_position_bias_table[idx.clamp(0, (2 * ws - 1) ** 2 - 1)].permute(2, 0, 1)      # [heads, L, L]
        logits = logits + bias.unsqueeze(0)
        logits = logits.masked_fill(~same_win, float("-inf"))
        logits = logits + torch.where(allowed, 0.0, -100.0).to(logits.dtype)                # the reference's additive mask
        out = m.proj((logits.softmax(-1) @ v).transpose(1, 2).reshape(2, Hp * Wp, C))
        ref = out.view(2, Hp, Wp, C)[:, :H, :W].reshape(2, H * W, C)
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("rows,C,G", [(1000, 96, 1), (333, 192, 1), (77, 768, 1), (250, 96, 4), (61, 384, 4)])
def test_layernorm_kernel_matches_float64(rows, C, G):
    from al3d import detector_ops as D
    from al3d import token_ops as T
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows * G, C, generator=g) * torch.exp(torch.randn(rows * G, 1, generator=g)) + 0.7).to(DEV)
    gamma = (torch.rand(G * C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(G * C, generator=g) * 0.1).to(DEV)
    ref = F.layer_norm(x.double().view(rows, G * C), (G * C,), gamma.double(), beta.double(), 1e-5)
    got = T.layernorm(x, gamma, beta, 1e-5, G=G)
    tol = 4e-6 * (1.0 + ref.abs())                     # a few fp32 roundings of the normalised value
    assert bool(((got.double() - ref).abs() <= tol).all())
    # pair rows hold the same values to 2^-22 relative (2^-36 absolute below f16's normal range)
    pair = D.rows_convert(T.layernorm(x, gamma, beta, 1e-5, G=G, pair=True), False)
    assert bool(((pair - got).abs() <= 2.0 ** -21 * got.abs() + 2.0 ** -35).all())


def test_layernorm_gather_maps():
    """Window order with zero OUTPUT rows for the padding (norm1 precedes the pad), and the 2 x 2 merge gather with zero
    INPUT pieces (the pad precedes the norm), against explicit torch indexing."""
    from al3d import token_ops as T
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 9, 11, 96
    x = torch.randn(B * H * W, C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    m, (nwy, nwx) = T.window_rowmap(B, H, W, 7, 3)
    assert (nwy, nwx) == (2, 2) and m.shape == (B * 4 * 49,) and sorted(m[m >= 0].tolist()) == list(range(B * H * W))
    mt = torch.from_numpy(m).to(DEV)
    got = T.layernorm(x, gamma, beta, 1e-5, rowmap=mt, zero_out=True)
    ln = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    ref = torch.where((mt >= 0)[:, None], ln[mt.clamp(min=0).long()], torch.zeros((), device=DEV))
    assert float((got - ref).abs().max()) <= 5e-6 * (1.0 + float(ref.abs().max())) and bool((got[mt < 0] == 0).all())
    # the window map agrees with roll + pad + partition of an index image
    idx = torch.arange(B * H * W, dtype=torch.float32).view(B, H, W, 1) + 1.0
    pad = F.pad(idx, (0, 0, 0, 14 - W, 0, 14 - H))
    sh = torch.roll(pad, (-3, -3), (1, 2)).view(B, 2, 7, 2, 7).permute(0, 1, 3, 2, 4).reshape(-1)
    assert torch.equal(sh.long() - 1, torch.from_numpy(m).long())
    # merge map: F.unfold of the padded index image, piece-major
    mm, (OH, OW) = T.merge_rowmap(B, H, W)
    un = F.unfold(F.pad(idx.permute(0, 3, 1, 2), (0, W % 2, 0, H % 2)), 2, stride=2)           # [B, 4, OH*OW], row kh*2+kw
    assert (OH, OW) == (5, 6) and torch.equal(un.transpose(1, 2).reshape(-1).long() - 1, torch.from_numpy(mm).long())
    g4, b4 = (torch.rand(4 * C, generator=g) + 0.5).to(DEV), torch.randn(4 * C, generator=g).to(DEV)
    mmt = torch.from_numpy(mm).to(DEV)
    got = T.layernorm(x, g4, b4, 1e-5, rowmap=mmt, G=4)
    pieces = torch.where((mmt >= 0)[:, None], x[mmt.clamp(min=0).long()], torch.zeros((), device=DEV)).view(-1, 4 * C)
    ref = F.layer_norm(pieces, (4 * C,), g4, b4, 1e-5)
    assert float((got - ref).abs().max()) <= 5e-6 * (1.0 + float(ref.abs().max()))


@pytest.mark.parametrize("M,K,N,xmag", [(1000, 96, 288, 1.0), (257, 96, 96, 1e-4), (128, 384, 96, 300.0),
                                        (300, 768, 3072, 1.0), (64, 3072, 768, 1.0), (5, 192, 576, 1.0)])
def test_token_gemm_is_fp32_class(M, K, N, xmag):
    """The f16x3 token GEMM against float64: <= 1.5e-6 of sum|a w| (the bound of test_conv2d_f16x3_is_fp32_class), and
    no further from the float64 value than 3x torch's own fp32 GEMM; f32 rows and pair rows in, the same bits."""
    from al3d import detector_ops as D
    from al3d import token_ops as T
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g)) * xmag).clamp(-6.0e4, 6.0e4).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    pk = T.PackedLinear(w, b)
    ref = a.double() @ w.double().t() + b.double()
    scale = a.abs().double() @ w.abs().double().t() + b.abs().double()
    got = T.linear(a, pk)
    e3 = float(((got.double() - ref).abs() / scale).max())
    e32 = float((((a @ w.t() + b).double() - ref).abs() / scale).max())
    assert e3 < 1.5e-6 and e3 < 3.0 * e32 + 1e-7, (e3, e32)
    a_pair = D.rows_convert(a, True)
    a_rt = D.rows_convert(a_pair, False)               # what the pair rows hold (22-23 significant bits)
    got_p = T.linear(a_pair, pk, a_pair=True)
    # pair rows in == their f32 values in (the same products, up to a re-split that may round xh the other way at a tie)
    ref_rt = a_rt.double() @ w.double().t() + b.double()
    assert float(((got_p.double() - ref_rt).abs() / scale).max()) < 1.5e-6
    assert float(((got_p - T.linear(a_rt, pk)).double().abs() / scale).max()) < 2e-7
    got_pp = D.rows_convert(T.linear(a_pair, pk, a_pair=True, out_pair=True), False) if N % 8 == 0 else None
    if got_pp is not None:
        assert bool(((got_pp - got_p).abs() <= 2.0 ** -21 * got_p.abs() + 2.0 ** -35).all())


def test_token_gemm_epilogues():
    """Exact GELU, residual, row scatter with dropped rows, in-place residual stream."""
    from al3d import token_ops as T
    g = torch.Generator().manual_seed(11)
    M, K, N, R = 300, 96, 96, 260
    a = torch.randn(M, K, generator=g).to(DEV)
    w, b = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
    pk = T.PackedLinear(w, b)
    lin = a.double() @ w.double().t() + b.double()
    got = T.linear(a, pk, act="gelu")
    assert float((got.double() - F.gelu(lin)).abs().max()) <= 3e-6
    perm = torch.randperm(M, generator=g)
    rowmap = torch.full((M,), -1, dtype=torch.int32)
    rowmap[perm[:R]] = torch.arange(R, dtype=torch.int32)          # R of the M rows land somewhere, the rest are dropped
    rowmap = rowmap.to(DEV)
    res = torch.randn(R, N, generator=g).to(DEV)
    stream = res.clone()
    out = T.linear(a, pk, residual=stream, rowmap=rowmap)          # scatter-add into the residual stream, in place
    assert out.data_ptr() == stream.data_ptr()
    ref = res.double().clone()
    ref[rowmap[rowmap >= 0].long()] += lin[(rowmap >= 0)]
    assert float((out.double() - ref).abs().max()) <= 3e-6 * (1.0 + float(ref.abs().max()))
    with pytest.raises(Exception):
        T.linear(a[:, :40].contiguous(), pk)                        # K mismatch


@pytest.mark.parametrize("B,H,W,heads,shift", [(2, 14, 21, 3, 0), (2, 16, 23, 3, 3), (1, 8, 22, 24, 3), (3, 7, 7, 6, 3)])
def test_window_attention_kernel_matches_float64(B, H, W, heads, shift):
    """qkv in window order -> attention, against the restatement's WindowMSA core in float64 (bias gather through
    mmdet's index, the additive 0 / -100 mask tensor) and bounded by torch's own fp32 evaluation."""
    import swin_torch as R
    from al3d import detector_ops as D
    from al3d import token_ops as T
    from al3d.models.swin import ShiftWindowMSA
    C = 32 * heads
    g = torch.Generator().manual_seed(H * W + heads)
    attn = ShiftWindowMSA(C, heads, 7, shift)
    attn.w_msa.relative_position_bias_table.data = torch.randn(169, heads, generator=g) * 0.7
    attn = attn.to(DEV).eval()
    m = attn.w_msa
    rowmap, (nwy, nwx) = T.window_rowmap(B, H, W, 7, shift)
    nwin = B * nwy * nwx
    qkv = (torch.randn(nwin * 49, 3 * C, generator=g) * 1.5).to(DEV)

    def core(dtype):
        x = qkv.to(dtype).view(nwin, 49, 3, heads, 32).permute(2, 0, 3, 1, 4)
        q, k, v = x[0], x[1], x[2]
        a = (q * m.scale) @ k.transpose(-2, -1)
        bias = m.relative_position_bias_table.to(dtype)[m.relative_position_index.view(-1)].view(49, 49, -1)
        a = a + bias.permute(2, 0, 1).unsqueeze(0)
        if shift:
            Hp, Wp = nwy * 7, nwx * 7
            img = torch.zeros((1, Hp, Wp, 1), device=DEV, dtype=dtype)
            cnt = 0
            for hs in (slice(0, -7), slice(-7, -shift), slice(-shift, None)):
                for ws_ in (slice(0, -7), slice(-7, -shift), slice(-shift, None)):
                    img[:, hs, ws_, :] = cnt
                    cnt += 1
            mw = R._window_partition(img, 7).view(-1, 49)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            am = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)
            a = (a.view(B, nwy * nwx, heads, 49, 49) + am.unsqueeze(1).unsqueeze(0)).view(-1, heads, 49, 49)
        return (a.softmax(-1) @ v).transpose(1, 2).reshape(nwin * 49, C)

    with torch.no_grad():
        ref, ref32 = core(torch.float64), core(torch.float32)
    got = T.window_attention(qkv, m.relative_position_bias_table.detach(), heads, nwy, nwx, shift, m.scale, pair=False)
    got = got.detach()
    e = float((got.double() - ref).abs().max())
    e32 = float((ref32.double() - ref).abs().max())
    assert e <= 3.0 * e32 + 2e-6 and e <= 1e-5, (e, e32)
    pair = D.rows_convert(T.window_attention(qkv, m.relative_position_bias_table.detach(), heads, nwy, nwx, shift, m.scale), False)
    assert bool(((pair - got).abs() <= 2.0 ** -21 * got.abs() + 2.0 ** -35).all())


# ------------------------------------------------------------------ modules
@pytest.mark.parametrize("B,H,W,C,heads,shift", [(2, 16, 23, 96, 3, True), (2, 14, 14, 192, 6, False), (1, 8, 22, 768, 24, True)])
def test_swin_block_matches_restatement(B, H, W, C, heads, shift):
    import swin_torch as R
    from al3d.models.swin import SwinBlock, _Geometry
    from al3d.synthetic import seed_modules_
    blk = seed_modules_(SwinBlock(C, heads, 4 * C, 7, shift), 17).to(DEV)
    x = torch.randn(B, H * W, C, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        ref = R.block(blk.double(), x.double(), (H, W))
        ref32 = R.block(blk.float(), x, (H, W))
        got = blk(x.clone().view(B * H * W, C), _Geometry.of(B, H, W, 7, x.device)).view(B, H * W, C)
    e, e32 = float((got.double() - ref).abs().max()), float((ref32.double() - ref).abs().max())
    assert e <= 3.0 * e32 + 1e-6 * float(ref.abs().max()), (e, e32)


def test_patch_merging_matches_restatement():
    import swin_torch as R
    from al3d.models.swin import PatchMerging, _Geometry
    from al3d.synthetic import seed_modules_
    B, H, W, C = 2, 9, 11, 96
    pm = seed_modules_(PatchMerging(C, 2 * C), 4).to(DEV)
    x = torch.randn(B, H * W, C, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        ref, hw = R.patch_merging(pm.double(), x.double(), (H, W))
        ref32, _ = R.patch_merging(pm.float(), x, (H, W))
        got, hw2 = pm(x.view(B * H * W, C), _Geometry.of(B, H, W, 7, x.device))
    assert hw == hw2 == (5, 6)
    e, e32 = float((got.view_as(ref).double() - ref).abs().max()), float((ref32.double() - ref).abs().max())
    assert e <= 3.0 * e32 + 1e-6 * float(ref.abs().max()), (e, e32)


def _swin_t():
    from al3d.models.swin import SwinTransformer
    from al3d.synthetic import seed_modules_
    return seed_modules_(SwinTransformer(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7,
                                         mlp_ratio=4, qkv_bias=True, patch_norm=True, out_indices=[1, 2, 3]), 23)


@pytest.mark.parametrize("B,H,W", [(2, 96, 160), (6, 256, 704)])
def test_swin_t_matches_restatement(B, H, W):
    """Whole backbone against the torch restatement in float64 (small) / float32 (full configs[4] size: 6 cameras of
    256 x 704), per output level, relative to the level's scale."""
    import swin_torch as R
    swin = _swin_t().to(DEV)
    img = torch.randn(B, H, W, 3, generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        got = swin(img)
        again = swin(img)
        ref32 = R.swin(swin, img)
        ref = R.swin(swin.double(), img, torch.float64) if H < 200 else None
        swin.float()
    assert all(torch.equal(a, b) for a, b in zip(got, again))                      # deterministic
    for lvl, (o, r32) in enumerate(zip(got, ref32)):
        assert o.shape == r32.shape and bool(torch.isfinite(o).all())
        s = float(r32.abs().max())
        if ref is not None:
            e, e32 = float((o.double() - ref[lvl]).abs().max()), float((r32.double() - ref[lvl]).abs().max())
            assert e <= 3.0 * e32 + 2e-6 * s, (lvl, e, e32, s)
        else:
            assert float((o - r32).abs().max()) <= 2e-4 * s, (lvl, float((o - r32).abs().max()), s)


def test_swin_t_levels_and_camera_branch_end_to_end():
    from al3d.models import DepthLSSTransform, GeneralizedLSSFPN
    from al3d.synthetic import camera_setup, seed_modules_
    torch.manual_seed(5)
    image_size, feature_size = (256, 704), (32, 88)
    swin = _swin_t().to(DEV)
    names = set(swin.state_dict())
    for key in ("patch_embed.projection.weight", "patch_embed.norm.weight", "stages.0.blocks.1.attn.w_msa.qkv.bias",
                "stages.2.blocks.5.attn.w_msa.relative_position_bias_table", "stages.1.blocks.0.ffn.layers.0.0.weight",
                "stages.1.blocks.0.ffn.layers.1.bias", "stages.0.downsample.reduction.weight", "stages.2.downsample.norm.bias",
                "norm1.weight", "norm3.bias"):
        assert key in names, key
    assert "norm0.weight" not in names and "stages.3.downsample.norm.weight" not in names
    B, N = 1, 6
    img = torch.randn(B * N, *image_size, 3, device=DEV)
    neck = seed_modules_(GeneralizedLSSFPN([192, 384, 768], 256, 3), 7).to(DEV)
    vt = seed_modules_(DepthLSSTransform(256, 80, image_size, feature_size, [-54.0, 54.0, 0.3], [-54.0, 54.0, 0.3],
                                         [-10.0, 10.0, 20.0], [1.0, 60.0, 0.5], downsample=2), 8).to(DEV)
    K, cam2lidar, lidar2image, img_aug, lidar_aug, points = camera_setup(B, N, 9, image_size)
    with torch.no_grad():
        feats = swin(img)
        assert [tuple(f.shape) for f in feats] == [(6, 32, 88, 192), (6, 16, 44, 384), (6, 8, 22, 768)]
        assert all(bool(torch.isfinite(f).all()) for f in feats)
        fpn = neck(list(feats))
        assert tuple(fpn[0].shape) == (6, 32, 88, 256)
        bev = vt(fpn[0].view(B, N, 32, 88, 256), [p.to(DEV) for p in points], lidar2image.to(DEV), K.to(DEV),
                 cam2lidar.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
    assert tuple(bev.shape) == (1, 180, 180, 80) and bool(torch.isfinite(bev).all()) and float(bev.abs().max()) > 0


def test_swin_reloads_weights_after_load_state_dict():
    """The packed-weight caches follow the parameters: a load_state_dict after the first forward changes
    the output accordingly."""
    import swin_torch as R
    from al3d.synthetic import seed_modules_
    swin = _swin_t().to(DEV)
    img = torch.randn(1, 64, 96, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        first = swin(img)
        other = {k: v.clone() for k, v in seed_modules_(_swin_t(), 99).state_dict().items()}
        swin.load_state_dict(other)
        second = swin(img)
        ref = R.swin(swin, img)
    assert not torch.equal(first[0], second[0])
    assert float((second[0] - ref[0]).abs().max()) <= 2e-4 * float(ref[0].abs().max())


def test_assembled_camera_lidar_model_runs_and_is_deterministic():
    """``BEVFusionCameraLidar`` (fusion_models/bevfusion.py:207-305 restated on this build's modules): one synthetic
    sample through camera encoder, lidar encoder, fuser, decoder, embedding tap and TransFusionHead; shapes, finiteness,
    run-to-run identical embedding, and the camera map's [x, y] -> [H=y, W=x] transposition (a camera-only change must
    move the embedding, a lidar-only change too)."""
    import os
    from al3d import synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames
    from al3d.models import build_detector
    from al3d.models.bevfusion_model import BEVFusionCameraLidar, transfusion_head_for
    from al3d.synthetic import camera_setup, seed_modules_
    from al3d.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py"))
    lidar = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(lidar, seed=0)
    model = BEVFusionCameraLidar(lidar, head=transfusion_head_for())
    for i, m in enumerate((model.camera_backbone, model.camera_neck, model.vtransform, model.fuser, model.head)):
        seed_modules_(m, 40 + i)
    model = model.to(DEV).eval()
    pool = PoolFrames.from_synthetic(2, DEV, num_base=2, seed=1)
    ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, None, 1, device=DEV)))
    K, cam2lidar, lidar2image, img_aug, lidar_aug, _ = camera_setup(1, 6, 9, (256, 704))
    img = torch.randn(1, 6, 256, 704, 3, generator=torch.Generator().manual_seed(2)).to(DEV)
    mats = (lidar2image.to(DEV), K.to(DEV), cam2lidar.to(DEV), img_aug.to(DEV), lidar_aug.to(DEV))
    with torch.no_grad():
        emb, dec, preds = model(ex, img, [pool.frames[0]], *mats)
        emb2, _, _ = model(ex, img, [pool.frames[0]], *mats)
        emb_cam, _, _ = model(ex, img.flip(3), [pool.frames[0]], *mats)           # mirrored images only
    assert tuple(emb.shape) == (1, 512) and tuple(dec.shape) == (1, 180, 180, 512) and bool(torch.isfinite(emb).all())
    assert torch.equal(emb, emb2)
    assert not torch.equal(emb, emb_cam)
    assert len(preds) == 1 and preds[0]["bboxes"].shape[1] == 9 and len(preds[0]["scores"]) > 0


@pytest.mark.parametrize("T_,C", [(1000, 96), (129, 96), (257, 96), (32, 96), (1000, 192), (193, 192), (385, 192), (16, 192), (5, 192)])
def test_fused_mlp_kernel_matches_float64_and_the_split_path(T_, C):
    """``al3d_tok_mlp_f16x3`` (LN2 + fc1 + exact GELU + fc2 + residual as one kernel, hidden activation in registers,
    everything transposed so an accumulator is the next product's operand) against the same formula in float64, at the
    bound the three-launch path meets -- it must not be further from float64 than 3x the split path (LN kernel + two token
    GEMMs), and within 2e-6 of the output scale; row counts that are not multiples of the 128-token tile."""
    from al3d import token_ops as Tk
    g = torch.Generator().manual_seed(T_ + C)
    x = (torch.randn(T_, C, generator=g) * 1.7 + 0.3)
    ln_w, ln_b = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
    w1, b1 = torch.randn(4 * C, C, generator=g) / C ** 0.5, torch.randn(4 * C, generator=g) * 0.1
    w2, b2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5, torch.randn(C, generator=g) * 0.1
    xd = x.double()
    xn = F.layer_norm(xd, (C,), ln_w.double(), ln_b.double(), 1e-5)
    ref = xd + F.gelu(xn @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    dev = lambda t: t.to(DEV)
    pk = Tk.PackedMlp(dev(ln_w), dev(ln_b), 1e-5, dev(w1), dev(b1), dev(w2), dev(b2))
    got = Tk.mlp(dev(x).clone(), pk).cpu().double()
    f1, f2 = Tk.PackedLinear(dev(w1), dev(b1)), Tk.PackedLinear(dev(w2), dev(b2))
    xs = dev(x).clone()
    hid = Tk.linear(Tk.layernorm(xs, dev(ln_w), dev(ln_b), 1e-5, pair=True), f1, a_pair=True, act="gelu", out_pair=True)
    split = Tk.linear(hid, f2, a_pair=True, residual=xs, out=xs).cpu().double()
    scale = float(ref.abs().max())
    e_fused, e_split = float((got - ref).abs().max()), float((split - ref).abs().max())
    print("fused", e_fused / scale, "split", e_split / scale)
    assert e_fused <= 3.0 * e_split + 1e-7 * scale, (e_fused, e_split, scale)
    assert e_fused <= 1e-5 * scale, (e_fused, scale)


@pytest.mark.parametrize("C,heads", [(96, 3), (192, 6)])
def test_swin_block_fused_and_split_mlp_agree(C, heads):
    """A whole block with the fused MLP (default at embed dims 96 / 192) against the same block with AL3D_SWIN_MLP=split
    semantics: agreement at fp32 rounding level of the residual stream."""
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    assert C in S.FUSED_MLP_DIMS
    blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, shift=True), 5).to(DEV)
    B, H, W = 2, 14, 21
    x = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(1)).to(DEV)
    geom = S._Geometry.of(B, H, W, 7, torch.device(DEV))
    saved = S.FUSED_MLP
    try:
        with torch.no_grad():
            S.FUSED_MLP = True
            a = blk(x.clone(), geom)
            S.FUSED_MLP = False
            b = blk(x.clone(), geom)
    finally:
        S.FUSED_MLP = saved
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_gelu_epilogue_accuracy():
    """The token GEMM's GELU epilogue (branch-free erf, csrc/tokens.hip ``tk_erf``) against float64 ``gelu`` on a dense
    grid of arguments over [-9, 9]: identity weights make the GEMM pass its input through (values of at most 11
    significant bits are exact in f16x3), so the error seen is the activation's own.  Bound: 2.5e-7 absolute for
    |v| <= 1 and 2.5e-7 relative to |v| beyond (fp32 rounding level); NaN / inf propagate."""
    from al3d import token_ops as Tk
    n = 1 << 16
    v = (torch.arange(n, dtype=torch.float64) / n * 18.0 - 9.0)
    v = (v * 64).round() / 64                                          # multiples of 2^-6 below 16: exact in f16
    K = 16
    a = torch.zeros(n, K)
    a[:, 0] = v.float()
    w = torch.zeros(8, K)
    w[0, 0] = 1.0
    pk = Tk.PackedLinear(w.to(DEV), torch.zeros(8, device=DEV))
    got = Tk.linear(a.to(DEV), pk, act="gelu").cpu()[:, 0].double()
    ref = F.gelu(v)
    err = (got - ref).abs()
    bound = 2.5e-7 * torch.clamp(v.abs(), min=1.0)
    assert bool((err <= bound).all()), (float((err / bound).max()), float(v[(err / bound).argmax()]))
    spec = torch.tensor([[float("nan")] + [0.0] * (K - 1), [float("inf")] + [0.0] * (K - 1), [-float("inf")] + [0.0] * (K - 1)])
    out = Tk.linear(spec.to(DEV), pk, act="gelu").cpu()[:, 0]
    assert torch.isnan(out[0]) and (out[1] == float("inf") or torch.isnan(out[1]))


@pytest.mark.parametrize("C", [96, 192])
def test_fused_mlp_edge_cases(C):
    """Empty input, a non-finite token (its row becomes NaN, the other rows of the same wave are untouched: tokens are the
    independent n dimension of every product) and the loud failures of the fused MLP entry point."""
    from al3d import lib, token_ops as Tk
    g = torch.Generator().manual_seed(9)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    pk = Tk.PackedMlp(mk(C) * 0.1 + 1, mk(C) * 0.1, 1e-5, mk(4 * C, C) / C ** 0.5, mk(4 * C) * 0.1, mk(C, 4 * C) / (4 * C) ** 0.5, mk(C) * 0.1)
    assert Tk.mlp(torch.empty(0, C, device=DEV), pk).shape == (0, C)
    x = mk(70, C)
    clean = Tk.mlp(x.clone(), pk)
    bad = x.clone()
    bad[5, 17] = float("nan")
    bad[40, 3] = float("inf")
    out = Tk.mlp(bad, pk)
    assert bool(torch.isnan(out[5]).all()) and not bool(torch.isfinite(out[40]).any())
    keep = [i for i in range(70) if i not in (5, 40)]
    assert torch.equal(out[keep], clean[keep])
    with pytest.raises(lib.Al3dError):
        Tk.PackedMlp(mk(384), mk(384), 1e-5, mk(1536, 384), mk(1536), mk(384, 1536), mk(384))    # built for C = 96 / 192
    with pytest.raises(lib.Al3dError):
        Tk.mlp(mk(8, 288 - C), pk)


def _attn_half_float64(x, hw, blk):
    """x + W-MSA(LN1(x)) in float64 through the restatement's own pieces (pad after the norm, cyclic shift, 7 x 7
    partition, region mask, relative position bias); the functions cast the module's parameters to x's dtype."""
    import swin_torch as R
    xd = x.double()
    return xd + R.shift_window_msa(blk.attn, R._ln(blk.norm1, xd), hw)


@pytest.mark.parametrize("B,H,W,C,heads,shift", [(2, 16, 23, 96, 3, 3), (2, 14, 21, 96, 3, 0), (2, 9, 16, 192, 6, 3),
                                                 (1, 7, 7, 192, 6, 0), (3, 20, 11, 96, 3, 3), (1, 7, 20, 96, 3, 3),
                                                 (1, 5, 6, 96, 3, 0)])
def test_fused_attention_half_matches_float64_and_the_split_path(B, H, W, C, heads, shift):
    """``al3d_tok_attn_block_f16x3`` (LN1 + qkv + 7 x 7 window attention + proj + residual as one kernel: two waves per
    (window, head), q / k / v / attention output never in memory) against the same half block in float64 and against the
    four-launch path (LayerNorm kernel, token GEMM, attention kernel, token GEMM with the scatter): not further from
    float64 than 3x the four launches, within 1e-5 of the output scale; maps that need padding, shifted and unshifted
    windows, both embed dims the kernel is built for, odd window counts (at embed dim 96 a workgroup holds two windows: the
    last one runs its second window dry) and a map smaller than one window."""
    from al3d import token_ops as Tk
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, shift > 0), 23 + C + shift).to(DEV)
    with torch.no_grad():                                  # LayerNorm parameters and biases away from their 1 / 0 defaults
        g = torch.Generator().manual_seed(3)
        blk.norm1.weight.copy_(torch.randn(C, generator=g) * 0.2 + 1.0)
        blk.norm1.bias.copy_(torch.randn(C, generator=g) * 0.1)
        blk.attn.w_msa.relative_position_bias_table.copy_(torch.randn(169, heads, generator=g) * 0.5)
    msa, n1 = blk.attn.w_msa, blk.norm1
    x = (torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(B + H)) * 1.3 + 0.2).to(DEV)
    geom = S._Geometry.of(B, H, W, 7, torch.device(DEV))
    rowmap, (nwy, nwx) = geom.window_map(blk.attn.shift_size)
    with torch.no_grad():
        ref = _attn_half_float64(x.view(B, H * W, C).cpu(), (H, W), copy.deepcopy(blk).cpu()).view(-1, C)
        fused = Tk.attn_block(x.clone(), B, H, W, msa.packed_fused(x.device, n1), blk.attn.shift_size, msa.scale)
        qkv_w, proj_w, table = msa.packed(x.device)
        xs = x.clone()
        xw = Tk.layernorm(xs, n1.weight, n1.bias, n1.eps, rowmap=rowmap, zero_out=True, pair=True)
        ao = Tk.window_attention(Tk.linear(xw, qkv_w, a_pair=True), table, heads, nwy, nwx, blk.attn.shift_size, msa.scale, pair=True)
        split = Tk.linear(ao, proj_w, a_pair=True, residual=xs, rowmap=rowmap, out=xs)
    scale = float(ref.abs().max())
    e_fused = float((fused.cpu().double() - ref).abs().max())
    e_split = float((split.cpu().double() - ref).abs().max())
    print("fused", e_fused / scale, "split", e_split / scale, "fused - split", float((fused - split).abs().max()) / scale)
    assert e_fused <= 3.0 * e_split + 1e-7 * scale, (e_fused, e_split, scale)
    assert e_fused <= 1e-5 * scale, (e_fused, scale)
    assert float((fused - split).abs().max()) <= 4e-6 * scale


def test_swin_block_fused_and_split_attention_agree():
    """Whole blocks (both embed dims) with the fused attention half (default) against AL3D_SWIN_ATTN=split semantics."""
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    for C, heads in ((96, 3), (192, 6)):
        blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, shift=True), 5).to(DEV)
        B, H, W = 2, 14, 21
        x = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(1)).to(DEV)
        geom = S._Geometry.of(B, H, W, 7, torch.device(DEV))
        saved = S.FUSED_ATTN
        try:
            with torch.no_grad():
                S.FUSED_ATTN = True
                a = blk(x.clone(), geom)
                S.FUSED_ATTN = False
                b = blk(x.clone(), geom)
        finally:
            S.FUSED_ATTN = saved
        assert float((a - b).abs().max()) <= 4e-6 * float(b.abs().max())


def test_fused_attention_half_edge_cases():
    """No window at all; a non-finite token poisons its own window only (windows are independent workgroups); loud
    failures for an embed dim the kernel is not built for and for a row count that does not match the maps."""
    from al3d import lib, token_ops as Tk
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    C, heads, B, H, W = 96, 3, 1, 14, 14
    blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, False), 2).to(DEV)
    pk = blk.attn.w_msa.packed_fused(torch.device(DEV), blk.norm1)
    assert Tk.attn_block(torch.empty(0, C, device=DEV), 0, H, W, pk, 0, 32 ** -0.5).shape == (0, C)
    x = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(4)).to(DEV)
    clean = Tk.attn_block(x.clone(), B, H, W, pk, 0, 32 ** -0.5)
    bad = x.clone()
    bad[3 * W + 2, 11] = float("nan")                       # token (3, 2): window (0, 0)
    out = Tk.attn_block(bad, B, H, W, pk, 0, 32 ** -0.5)
    win00 = torch.zeros(H, W, dtype=torch.bool)
    win00[:7, :7] = True
    win00 = win00.view(-1).to(DEV)
    assert bool(torch.isnan(out[win00]).all()) and torch.equal(out[~win00], clean[~win00])
    with pytest.raises(lib.Al3dError):
        seed = seed_modules_(S.SwinBlock(384, 12, 1536, 7, False), 2).to(DEV)
        seed.attn.w_msa.packed_fused(torch.device(DEV), seed.norm1)
    with pytest.raises(lib.Al3dError):
        Tk.attn_block(x.clone(), B, H, W + 1, pk, 0, 32 ** -0.5)


@pytest.mark.parametrize("B,H,W,C,heads,shift", [(2, 16, 23, 384, 12, True), (1, 8, 22, 768, 24, True), (2, 14, 21, 384, 12, False),
                                                 (1, 5, 6, 384, 12, True)])
def test_swin_block_token_order_equals_window_order(B, H, W, C, heads, shift):
    """The unfused attention half in token order (LN1 / qkv / proj on the map's tokens, ``al3d_tok_window_attention_tokens_f32``
    gathers the shifted, padded windows and writes token order back; a padded position's q / k / v is the qkv bias) against
    the window-order path (LN1 gathers through the row map, zero rows for padding, the projection scatters back): every GEMM
    row and every window sees the same numbers, so the residual stream is bit-identical."""
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, shift), 31).to(DEV)
    x = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(C + H)).to(DEV)
    geom = S._Geometry.of(B, H, W, 7, torch.device(DEV))
    saved = S.TOKEN_ORDER
    try:
        with torch.no_grad():
            S.TOKEN_ORDER = True
            a = blk(x.clone(), geom)
            S.TOKEN_ORDER = False
            b = blk(x.clone(), geom)
    finally:
        S.TOKEN_ORDER = saved
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("C,heads", [(96, 3), (384, 12)])
def test_swin_block_without_qkv_bias(C, heads):
    """``qkv_bias=False``: the fused attention half (embed dim 96) gets a zero bias vector, the token-order attention
    (embed dim 384) reads a zero row for padded window positions -- both against the window-order four-launch path."""
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    blk = seed_modules_(S.SwinBlock(C, heads, 4 * C, 7, shift=True, qkv_bias=False), 11).to(DEV)
    assert blk.attn.w_msa.qkv.bias is None
    B, H, W = 2, 9, 16
    x = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(7)).to(DEV)
    geom = S._Geometry.of(B, H, W, 7, torch.device(DEV))
    saved = (S.FUSED_ATTN, S.TOKEN_ORDER)
    try:
        with torch.no_grad():
            a = blk(x.clone(), geom)
            S.FUSED_ATTN, S.TOKEN_ORDER = False, False
            b = blk(x.clone(), geom)
    finally:
        S.FUSED_ATTN, S.TOKEN_ORDER = saved
    assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= 4e-6 * float(b.abs().max())


@pytest.mark.parametrize("B,H,W,norm", [(2, 64, 96, True), (1, 30, 44, True), (3, 9, 8, False), (1, 256, 704, True)])
def test_fused_patch_embedding_matches_float64_and_the_three_launches(B, H, W, norm):
    """``al3d_tok_patch_embed_f16x3`` (4 x 4 / stride-4 projection + LayerNorm as one kernel) against conv2d + layer_norm in
    float64 and against the three-launch path (patch rows, token GEMM, LayerNorm kernel): heights that are not multiples of
    4 (zero rows below the image), with and without the norm, the configs[4] image size."""
    from al3d.models import swin as S
    from al3d.synthetic import seed_modules_
    pe = seed_modules_(S._PatchEmbed(3, 96, 4, norm), 13).to(DEV)
    if norm:
        with torch.no_grad():
            g = torch.Generator().manual_seed(5)
            pe.norm.weight.copy_(torch.randn(96, generator=g) * 0.2 + 1.0)
            pe.norm.bias.copy_(torch.randn(96, generator=g) * 0.1)
    img = (torch.randn(B, H, W, 3, generator=torch.Generator().manual_seed(H + W)) * 1.2 + 0.1).to(DEV)
    with torch.no_grad():
        xd = img.double().permute(0, 3, 1, 2)
        xd = F.pad(xd, (0, 0, 0, (-H) % 4))
        ref = F.conv2d(xd, pe.projection.weight.double(), pe.projection.bias.double(), stride=4).flatten(2).transpose(1, 2)
        if norm:
            ref = F.layer_norm(ref, (96,), pe.norm.weight.double(), pe.norm.bias.double(), pe.norm.eps)
        ref = ref.reshape(-1, 96)
        saved = S.FUSED_PATCH_EMBED
        try:
            S.FUSED_PATCH_EMBED = True
            fused, hw = pe(img)
            S.FUSED_PATCH_EMBED = False
            split, hw2 = pe(img)
        finally:
            S.FUSED_PATCH_EMBED = saved
    assert hw == hw2 == ((H + 3) // 4, W // 4) and fused.shape == split.shape == ref.shape
    scale = float(ref.abs().max())
    e_f, e_s = float((fused.double() - ref).abs().max()), float((split.double() - ref).abs().max())
    print("fused", e_f / scale, "split", e_s / scale)
    assert e_f <= 3.0 * e_s + 2e-7 * scale and e_f <= 2e-6 * scale, (e_f, e_s, scale)
