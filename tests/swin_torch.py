"""Torch (library-op) restatement of the Swin-T forward pass, used ONLY as the checker of the HIP token kernels
(``al3d/models/swin.py`` runs every stage on ``csrc/tokens.hip``).  It evaluates the published algorithm (Liu et al.,
ICCV 2021) in mmdet 2.20.0's conventions -- as far as its checkpoints' key names show them; mmdet is not in the reference
tree, parity unpinned -- with the PRODUCT modules' parameters: nn.functional.layer_norm / linear / softmax, torch.roll,
explicit window partition, the additive 0 / -100 region mask, ``nn.Unfold`` channel order in patch merging.  ``dtype``
float64 gives the high-precision evaluation the fp32-class bound is measured against."""
import torch
import torch.nn.functional as F


def _lin(m, x):
    return F.linear(x, m.weight.to(x.dtype), None if m.bias is None else m.bias.to(x.dtype))


def _ln(m, x):
    return F.layer_norm(x, m.normalized_shape, m.weight.to(x.dtype), m.bias.to(x.dtype), m.eps)


def window_msa(m, x, mask=None):
    """x [num_windows * B, N, C]; mask [num_windows, N, N] (0 / -100) or None."""
    B, N, C = x.shape
    qkv = _lin(m.qkv, x).reshape(B, N, 3, m.num_heads, C // m.num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * m.scale) @ k.transpose(-2, -1)
    bias = m.relative_position_bias_table.to(x.dtype)[m.relative_position_index.view(-1)].view(N, N, -1)
    attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B // nW, nW, m.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, m.num_heads, N, N)
    attn = attn.softmax(dim=-1)
    return _lin(m.proj, (attn @ v).transpose(1, 2).reshape(B, N, C))


def _window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def _window_reverse(windows, H, W, ws):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_window_msa(m, query, hw_shape):
    B, L, C = query.shape
    H, W = hw_shape
    assert L == H * W
    ws, shift = m.window_size, m.shift_size
    query = query.view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    query = F.pad(query, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = query.shape[1], query.shape[2]
    if shift > 0:
        shifted = torch.roll(query, shifts=(-shift, -shift), dims=(1, 2))
        img_mask = torch.zeros((1, Hp, Wp, 1), device=query.device, dtype=query.dtype)
        slices = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
        cnt = 0
        for h in slices:
            for w in slices:
                img_mask[:, h, w, :] = cnt
                cnt += 1
        mask_windows = _window_partition(img_mask, ws).view(-1, ws * ws)
        attn_mask = mask_windows.unsqueeze(1) - mask_windows.unsqueeze(2)
        attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
    else:
        shifted, attn_mask = query, None
    windows = _window_partition(shifted, ws).view(-1, ws * ws, C)
    attn_windows = window_msa(m.w_msa, windows, mask=attn_mask).view(-1, ws, ws, C)
    shifted = _window_reverse(attn_windows, Hp, Wp, ws)
    x = torch.roll(shifted, shifts=(shift, shift), dims=(1, 2)) if shift > 0 else shifted
    if pad_r > 0 or pad_b:
        x = x[:, :H, :W, :].contiguous()
    return x.view(B, H * W, C)


def ffn(m, x):
    return _lin(m.layers[1], F.gelu(_lin(m.layers[0][0], x)))


def block(blk, x, hw_shape):
    x = x + shift_window_msa(blk.attn, _ln(blk.norm1, x), hw_shape)
    return x + ffn(blk.ffn, _ln(blk.norm2, x))


def patch_merging(pm, x, hw_shape):
    B, L, C = x.shape
    H, W = hw_shape
    x = x.view(B, H, W, C).permute(0, 3, 1, 2)
    x = F.pad(x, (0, W % 2, 0, H % 2))                                    # "corner" padding to even sizes
    x = F.unfold(x, kernel_size=2, stride=2)                              # [B, C*4, L/4], index c*4 + kh*2 + kw
    out_hw = ((H + 1) // 2, (W + 1) // 2)
    return _lin(pm.reduction, _ln(pm.norm, x.transpose(1, 2))), out_hw


def patch_embed(pe, img):
    """img channels-last [B,H,W,3] -> tokens [B, L, C], (H/4, W/4)."""
    ps = pe.patch_size
    B, H, W, _ = img.shape
    x = F.pad(img.permute(0, 3, 1, 2), (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
    y = F.conv2d(x, pe.projection.weight.to(x.dtype), pe.projection.bias.to(x.dtype), stride=ps)
    hw = (y.shape[2], y.shape[3])
    y = y.flatten(2).transpose(1, 2)
    return (_ln(pe.norm, y) if pe.norm is not None else y), hw


def swin(model, img, dtype=torch.float32):
    """The whole backbone: tuple of channels-last maps [B, H_l, W_l, C_l] for ``model.out_indices``."""
    x, hw = patch_embed(model.patch_embed, img.to(dtype))
    outs = []
    for i, stage in enumerate(model.stages):
        for blk in stage.blocks:
            x = block(blk, x, hw)
        out, out_hw = x, hw
        if stage.downsample is not None:
            x, hw = patch_merging(stage.downsample, x, hw)
        if i in model.out_indices:
            o = _ln(getattr(model, f"norm{i}"), out)
            outs.append(o.view(-1, *out_hw, model.num_features[i]).contiguous())
    return tuple(outs)
