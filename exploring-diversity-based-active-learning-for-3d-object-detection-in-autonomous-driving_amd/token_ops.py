"""Device-side token-matrix primitives of the Swin-T image backbone (csrc/tokens.hip behind include/al3d.h's
``al3d_tok_*``): LayerNorm over gathered rows, the f16x3 token GEMM with its fused epilogues, and 7 x 7 window
attention.  torch tensors in, HIP kernels underneath; no CPU fallback.

Activations are row matrices ``[rows, C]`` float32; "pair rows" (``pair=True``) hold the same bytes per row as the
two f16 planes the f16x3 products multiply with (csrc/sp_rows.h) and only ever travel from a producing kernel to the
GEMM that consumes them.
"""
import numpy as np
import torch

from . import lib
from .detector_ops import pack_dma_f16x3, split_f16x3
from .selector_ops import _dev, _ptr, _stream


class PackedLinear:
    """``nn.Linear`` / 1x1 ``Conv1d`` weights in the token GEMM's format: LDS-DMA image of the (wh, wl) f16 planes, the
    2^-s scale of the split (times ``scale``, e.g. a folded BatchNorm), the bias / shift.  ``pad_k`` / ``pad_n``
    zero-pad the input / output channels (K must be a multiple of 16, N of 4)."""

    def __init__(self, weight, bias=None, scale=None, pad_k=None, pad_n=None):
        w = weight.detach().float().reshape(weight.shape[0], -1)
        n, k = w.shape
        pad_k, pad_n = pad_k or k, pad_n or n
        if pad_k != k or pad_n != n:
            w = torch.nn.functional.pad(w, (0, pad_k - k, 0, pad_n - n))
        w = w.contiguous()
        self.n, self.k = w.shape
        if self.k % 16:
            raise lib.Al3dError(f"PackedLinear: in_features={self.k} must be a multiple of 16 (use pad_k)")
        if scale is not None:
            scale = torch.nn.functional.pad(scale.detach().float(), (0, pad_n - n)).to(w.device)
        planes, sc = split_f16x3(w.view(self.n, 1, self.k), scale)
        self.image = pack_dma_f16x3(planes).data
        self.scale = sc
        self.bias = None
        if bias is not None:
            self.bias = torch.nn.functional.pad(bias.detach().float(), (0, pad_n - n)).to(w.device).contiguous()


class PackedMlp:
    """fc1 / fc2 of a Swin block in the fused MLP kernel's format (``al3d_tok_mlp_f16x3``): per 32 hidden units the MFMA
    A-operand fragments of both matrices in lane order, (wh, wl) planes of the f16x3 split, one power-of-two scale per
    matrix; LayerNorm parameters and biases as f32 vectors."""

    def __init__(self, norm_weight, norm_bias, eps, fc1_weight, fc1_bias, fc2_weight, fc2_bias):
        w1 = fc1_weight.detach().float().contiguous()            # [hidden, C]
        w2 = fc2_weight.detach().float().contiguous()            # [C, hidden]
        hidden, C = w1.shape
        if C not in (96, 192) or hidden % 32 or tuple(w2.shape) != (C, hidden):
            raise lib.Al3dError(f"PackedMlp: C={C} (the fused kernel is built for 96 / 192), hidden={hidden} (multiple of 32), fc2 {tuple(w2.shape)}")
        p1, s1 = split_f16x3(w1.view(hidden, 1, C))
        p2, s2 = split_f16x3(w2.view(C, 1, hidden))
        NT = hidden // 32
        if C == 96:
            KC, U = C // 16, C // 32
            # fc1: [plane, t, fr, kc, fh, e] -> [t, kc, plane, fh, fr, e]  (lane = fh * 32 + fr)
            a = p1.view(2, NT, 32, KC, 2, 8).permute(1, 3, 0, 4, 2, 5).reshape(NT, KC * 2 * 64 * 8)
            # fc2: hidden unit inside tile = 16 q + 8 e_hi + 4 fh + e_lo: [plane, u, fr, t, q, e_hi, fh, e_lo] -> [t, u, q, plane, fh, fr, e_hi, e_lo]
            b = p2.view(2, U, 32, NT, 2, 2, 2, 4).permute(3, 1, 4, 0, 6, 2, 5, 7).reshape(NT, U * 2 * 2 * 64 * 8)
        else:
            # 16 x 16 x 32 fragments (tok_mlp16_f16x3_kernel): lane = lq * 16 + lr holds k-block lq of row lr
            KS, U = C // 32, C // 16
            # fc1: [plane, g, j, lr, ks, lq, e] -> [g, j, ks, plane, lq, lr, e]
            a = p1.view(2, NT, 2, 16, KS, 4, 8).permute(1, 2, 4, 0, 5, 3, 6).reshape(NT, 2 * KS * 2 * 64 * 8)
            # fc2: the k order of the accumulator-made B fragment: (lq, e) -> unit 4 lq + e (e < 4), 16 + 4 lq + e - 4 (e >= 4)
            lq = torch.arange(4).view(4, 1)
            e = torch.arange(8).view(1, 8)
            perm = torch.where(e < 4, 4 * lq + e, 16 + 4 * lq + e - 4).reshape(32).to(p2.device)
            # [plane, u, lr, g, unit] -> units permuted -> [plane, u, lr, g, lq, e] -> [g, u, plane, lq, lr, e]
            b = p2.view(2, U, 16, NT, 32)[..., perm].view(2, U, 16, NT, 4, 8).permute(3, 1, 0, 4, 2, 5).reshape(NT, U * 2 * 64 * 8)
        self.image = torch.cat([a, b], dim=1).contiguous()
        assert self.image.numel() * 2 == lib.load().al3d_tok_mlp_image_bytes(C, hidden)
        self.C, self.hidden, self.eps = C, hidden, float(eps)
        self.scale1, self.scale2 = float(s1[0]), float(s2[0])
        dev = self.image.device
        f = lambda v, n: (torch.zeros(n) if v is None else v.detach().float()).to(dev).contiguous()
        self.gamma, self.beta = f(norm_weight, C), f(norm_bias, C)
        self.bias1, self.bias2 = f(fc1_bias, hidden), f(fc2_bias, C)


class PackedAttnBlock:
    """LN1 + qkv + proj of a Swin block in the fused attention kernel's format (``al3d_tok_attn_block_f16x3``): per head
    the MFMA fragments of its k, v, q rows of ``qkv.weight`` and of rows 32 h .. 32 h + 31 of ``proj.weight`` in lane
    order, (wh, wl) planes, one power-of-two scale per matrix; LayerNorm parameters, biases and the relative position
    bias table as f32 tensors."""

    def __init__(self, norm_weight, norm_bias, eps, qkv_weight, qkv_bias, proj_weight, proj_bias, table):
        wq = qkv_weight.detach().float().contiguous()            # [3C, C]: q | k | v rows
        wp = proj_weight.detach().float().contiguous()           # [C, C]
        C = wp.shape[0]
        if C not in (96, 192) or tuple(wq.shape) != (3 * C, C) or tuple(wp.shape) != (C, C):
            raise lib.Al3dError(f"PackedAttnBlock: C={C} (the fused kernel is built for 96 / 192), qkv {tuple(wq.shape)}")
        H, KC = C // 32, C // 16
        pq, sq = split_f16x3(wq.view(3 * C, 1, C))
        pp, sp = split_f16x3(wp.view(C, 1, C))
        # [plane, which, h, fr, kc, fh, e] -> [h, which (k, v, q), kc, plane, fh, fr, e]   (lane = fh * 32 + fr)
        a = pq.view(2, 3, H, 32, KC, 2, 8)[:, [1, 2, 0]].permute(2, 1, 4, 0, 5, 3, 6)
        b = pp.view(2, 1, H, 32, KC, 2, 8).permute(2, 1, 4, 0, 5, 3, 6)
        self.image = torch.cat([a, b], dim=1).contiguous()
        assert self.image.numel() * 2 == lib.load().al3d_tok_attn_block_image_bytes(C)
        self.C, self.heads, self.eps = C, H, float(eps)
        self.scale_qkv, self.scale_proj = float(sq[0]), float(sp[0])
        dev = self.image.device
        f = lambda v, n: (torch.zeros(n) if v is None else v.detach().float()).to(dev).contiguous()
        self.gamma, self.beta = f(norm_weight, C), f(norm_bias, C)
        self.bias_qkv, self.bias_proj = f(qkv_bias, 3 * C), f(proj_bias, C)
        self.table = table.detach().float().to(dev).contiguous()
        if tuple(self.table.shape) != (169, H):
            raise lib.Al3dError(f"PackedAttnBlock: relative position bias table {tuple(self.table.shape)}, expected (169, {H})")


def attn_block(x, B, H, W, packed, shift, scale):
    """``x += proj(shifted_window_attention(LN(x)))`` in place on the f32 token rows ``[B * H * W, C]`` of ``B`` maps
    (7 x 7 windows, cyclic shift ``shift``, padding to multiples of 7 after the norm; one launch)."""
    x = _dev(x, torch.float32, "x")
    if x.dim() != 2 or x.shape[-1] != packed.C:
        raise lib.Al3dError(f"attn_block: x {tuple(x.shape)}, the weights have {packed.C} channels")
    if x.shape[0] != B * H * W:
        raise lib.Al3dError(f"attn_block: x has {x.shape[0]} rows, {B} maps of {H} x {W} need {B * H * W}")
    lib.call("al3d_tok_attn_block_f16x3", _ptr(x), B, H, W, packed.C, int(shift), _ptr(packed.gamma), _ptr(packed.beta),
             packed.eps, _ptr(packed.image), packed.scale_qkv, _ptr(packed.bias_qkv), packed.scale_proj,
             _ptr(packed.bias_proj), _ptr(packed.table), float(scale), _stream())
    return x


def mlp(x, packed):
    """``x += fc2(gelu(fc1(LN(x))))`` in place on f32 token rows ``[T, 96]`` (one launch)."""
    x = _dev(x, torch.float32, "x")
    if x.shape[-1] != packed.C:
        raise lib.Al3dError(f"mlp: x has {x.shape[-1]} channels, the weights {packed.C}")
    lib.call("al3d_tok_mlp_f16x3", _ptr(x), x.shape[0], packed.C, packed.hidden, _ptr(packed.gamma), _ptr(packed.beta),
             packed.eps, _ptr(packed.image), packed.scale1, _ptr(packed.bias1), packed.scale2, _ptr(packed.bias2), _stream())
    return x


class PackedPatchEmbed:
    """Conv2d(3, 96, 4, stride 4) (+ LayerNorm(96)) in the fused patch-embedding kernel's format
    (``al3d_tok_patch_embed_f16x3``): the MFMA A fragments of the projection weight reordered to k = (ky, kx, c)."""

    def __init__(self, weight, bias, norm_weight=None, norm_bias=None, eps=1e-5):
        w = weight.detach().float()
        if tuple(w.shape) != (96, 3, 4, 4):
            raise lib.Al3dError(f"PackedPatchEmbed: weight {tuple(w.shape)}, the fused kernel is built for (96, 3, 4, 4)")
        w2 = w.permute(0, 2, 3, 1).reshape(96, 48).contiguous()
        planes, sc = split_f16x3(w2.view(96, 1, 48))
        # [plane, u, fr, s, fh, e] -> [u, s, plane, fh, fr, e]   (lane = fh * 32 + fr)
        self.image = planes.view(2, 3, 32, 3, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
        assert self.image.numel() * 2 == lib.load().al3d_tok_patch_embed_image_bytes()
        self.scale, self.eps = float(sc[0]), float(eps)
        dev = self.image.device
        f = lambda v: None if v is None else v.detach().float().to(dev).contiguous()
        self.bias, self.gamma, self.beta = f(bias), f(norm_weight), f(norm_bias)


def patch_embed(img, packed):
    """img ``[B, H, W, 3]`` channels-last (W a multiple of 4) -> token rows ``[B * ceil(H/4) * (W/4), 96]`` f32: projection
    (+ LayerNorm) in one launch."""
    img = _dev(img, torch.float32, "img")
    B, H, W, ch = img.shape
    if ch != 3 or W % 4:
        raise lib.Al3dError("patch_embed: expected 3 image channels and a width that is a multiple of 4")
    th, tw = (H + 3) // 4, W // 4
    out = torch.empty((B * th * tw, 96), dtype=torch.float32, device=img.device)
    lib.call("al3d_tok_patch_embed_f16x3", _ptr(img), B, H, W, _ptr(packed.image), packed.scale, _ptr(packed.bias),
             _ptr(packed.gamma), _ptr(packed.beta), packed.eps, _ptr(out), _stream())
    return out, (th, tw)


def patch_rows(img, pair=True):
    """img ``[B, H, W, 3]`` channels-last -> 4 x 4 patch rows ``[B * ceil(H/4) * ceil(W/4), 48]`` (k = (ky*4 + kx)*3 + c)."""
    img = _dev(img, torch.float32, "img")
    B, H, W, ch = img.shape
    if ch != 3:
        raise lib.Al3dError("patch_rows: expected 3 image channels")
    th, tw = (H + 3) // 4, (W + 3) // 4
    out = torch.empty((B * th * tw, 48), dtype=torch.float32, device=img.device)
    lib.call("al3d_tok_patch_rows_f32", _ptr(img), B, H, W, int(pair), _ptr(out), _stream())
    return out, (th, tw)


def layernorm(x, gamma, beta, eps, rowmap=None, G=1, zero_out=False, rows_out=None, pair=False):
    """LN over ``G * C`` channels of (gathered) rows of ``x [rows, C]`` -> ``[rows_out, G * C]``; see
    ``al3d_tok_layernorm_f32``."""
    x = _dev(x, torch.float32, "x")
    C = x.shape[-1]
    if rowmap is None:
        rows_out = x.shape[0] // G if rows_out is None else rows_out
    else:
        rowmap = _dev(rowmap, torch.int32, "rowmap")
        rows_out = rowmap.numel() // G
    out = torch.empty((rows_out, G * C), dtype=torch.float32, device=x.device)
    lib.call("al3d_tok_layernorm_f32", _ptr(x), _ptr(rowmap), rows_out, C, G, int(zero_out),
             _ptr(_dev(gamma, torch.float32, "gamma")), _ptr(_dev(beta, torch.float32, "beta")), float(eps), int(pair),
             _ptr(out), _stream())
    return out


def linear(a, packed, a_pair=False, act=None, residual=None, rowmap=None, out=None, out_rows=None, out_pair=False):
    """``out[rowmap[m]] = act(a[m] @ W^T + b) + residual[rowmap[m]]``; ``residual`` may be ``out`` itself."""
    a = _dev(a, torch.float32, "a")
    M, K = a.shape
    if K != packed.k:
        raise lib.Al3dError(f"linear: a has {K} columns, the weights {packed.k}")
    if rowmap is not None:
        rowmap = _dev(rowmap, torch.int32, "rowmap")
        if rowmap.numel() != M:
            raise lib.Al3dError("linear: rowmap needs one entry per row of a")
    if out is None:
        if residual is not None and rowmap is not None:
            out = residual                                   # scatter-add back into the residual stream, in place
        else:
            # with a row map the kernel skips dropped rows (-1) and may leave rows of a larger ``out_rows`` untouched: those
            # must read as zeros, not as whatever the allocator held
            alloc = torch.zeros if rowmap is not None else torch.empty
            out = alloc((M if out_rows is None else out_rows, packed.n), dtype=torch.float32, device=a.device)
    if residual is not None:
        residual = _dev(residual, torch.float32, "residual")
    lib.call("al3d_tok_linear_f16x3", _ptr(a), int(a_pair), _ptr(packed.image), _ptr(packed.scale), _ptr(packed.bias),
             M, K, packed.n, {None: 0, "gelu": 1, "relu": 2}[act], _ptr(residual), 0 if residual is None else residual.shape[-1],
             _ptr(rowmap), _ptr(out), out.shape[-1], int(out_pair), _stream())
    return out


def window_attention(qkv, table, heads, win_rows, win_cols, shift, scale, pair=True):
    """qkv ``[nwin * 49, 3 C]`` in window order -> attention output ``[nwin * 49, C]`` (pair rows by default)."""
    qkv = _dev(qkv, torch.float32, "qkv")
    rows, c3 = qkv.shape
    C = c3 // 3
    nwin = rows // 49
    out = torch.empty((rows, C), dtype=torch.float32, device=qkv.device)
    lib.call("al3d_tok_window_attention_f32", _ptr(qkv), _ptr(_dev(table, torch.float32, "table")), nwin, C, heads,
             win_rows, win_cols, int(shift), float(scale), int(pair), _ptr(out), _stream())
    return out


def window_attention_tokens(qkv, bias_qkv, table, B, H, W, heads, shift, scale, pair=True):
    """qkv ``[B * H * W, 3 C]`` in TOKEN order -> attention output ``[B * H * W, C]`` in token order (pair rows by default):
    shift, padding and window partition from the window's position; a padded position's q / k / v is ``bias_qkv``."""
    qkv = _dev(qkv, torch.float32, "qkv")
    rows, c3 = qkv.shape
    C = c3 // 3
    if rows != B * H * W:
        raise lib.Al3dError(f"window_attention_tokens: qkv has {rows} rows, {B} maps of {H} x {W} need {B * H * W}")
    out = torch.empty((rows, C), dtype=torch.float32, device=qkv.device)
    lib.call("al3d_tok_window_attention_tokens_f32", _ptr(qkv), _ptr(_dev(bias_qkv, torch.float32, "bias_qkv")),
             _ptr(_dev(table, torch.float32, "table")), B, H, W, C, heads, int(shift), float(scale), int(pair), _ptr(out), _stream())
    return out


def mha16(q, k, v, B, Pq, Pk, heads, scale):
    """Attention core of ``nn.MultiheadAttention`` for 16-channel heads: q ``[B * Pq, >= heads * 16]``, k / v
    ``[B * Pk, ...]`` (column slices of wider row matrices are fine: the row pitch is the tensor's stride) ->
    ``[B * Pq, heads * 16]``."""
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise lib.Al3dError(f"mha16: {name} must be a device float32 row matrix (unit column stride)")
    if q.shape[0] != B * Pq or k.shape[0] != B * Pk or v.shape[0] != B * Pk:
        raise lib.Al3dError("mha16: row counts do not match B, Pq, Pk")
    out = torch.empty((B * Pq, heads * 16), dtype=torch.float32, device=q.device)
    ws = torch.empty(lib.load().al3d_tok_mha16_workspace_bytes(B, heads, Pq, Pk), dtype=torch.uint8, device=q.device)
    lib.call("al3d_tok_mha16_f32", q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), B, heads,
             Pq, Pk, float(scale), _ptr(out), out.shape[1], _ptr(ws), _stream())
    return out


def window_rowmap(B, H, W, ws, shift):
    """Row map of the (shifted) window partition of ``B`` maps of ``H x W`` tokens: entry ``((b * nwy + wy) * nwx + wx) *
    ws^2 + ty * ws + tx`` = index of the token that the cyclic shift by ``-shift`` and the padding to multiples of
    ``ws`` put at that window position, or -1 for padding.  The same map scatters the attention output back
    (``window_reverse`` + the inverse roll + the crop).  numpy int32 + (nwy, nwx)."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    nwy, nwx = Hp // ws, Wp // ws
    hs = (np.arange(Hp) + shift) % Hp                     # shifted[h] = padded[(h + shift) % Hp]
    wsf = (np.arange(Wp) + shift) % Wp
    src = np.where((hs[:, None] < H) & (wsf[None, :] < W), hs[:, None] * W + wsf[None, :], -1)      # [Hp, Wp]
    src = src.reshape(nwy, ws, nwx, ws).transpose(0, 2, 1, 3).reshape(-1)
    full = np.where(src[None, :] >= 0, src[None, :] + (np.arange(B) * H * W)[:, None], -1)
    return full.reshape(-1).astype(np.int32), (nwy, nwx)


def merge_rowmap(B, H, W):
    """Row map of patch merging: output token (b, oy, ox), piece kh * 2 + kw <- token (2 oy + kh, 2 ox + kw), -1
    where the map is padded to even sizes.  numpy int32 [B * OH * OW * 4] + (OH, OW)."""
    OH, OW = (H + 1) // 2, (W + 1) // 2
    oy, ox, kh, kw = np.meshgrid(np.arange(OH), np.arange(OW), np.arange(2), np.arange(2), indexing="ij")
    y, x = 2 * oy + kh, 2 * ox + kw
    src = np.where((y < H) & (x < W), y * W + x, -1).reshape(-1)
    full = np.where(src[None, :] >= 0, src[None, :] + (np.arange(B) * H * W)[:, None], -1)
    return full.reshape(-1).astype(np.int32), (OH, OW)
