"""Swin Transformer image backbone of BEVFusion's camera branch (BASELINE configs[4]).

The reference configures ``type: SwinTransformer`` from **mmdet 2.20.0** (``mmdet/models/backbones/swin.py``;
bevfusion/configs/nuscenes/det/transfusion/secfpn/camera+lidar/swint_v0p075/default.yaml: embed_dims 96, depths
[2, 2, 6, 2], num_heads [3, 6, 12, 24], window_size 7, mlp_ratio 4, qkv_bias, patch_norm, out_indices [1, 2, 3]).
mmdet is not in the reference tree and not installed, so this restates the PUBLISHED algorithm (Liu et al., "Swin
Transformer", ICCV 2021) with mmdet's module tree and conventions where they are observable from its checkpoints'
key names: ``patch_embed.projection`` / ``.norm``; ``stages.<i>.blocks.<j>.{norm1, attn.w_msa.{qkv, proj,
relative_position_bias_table}, norm2, ffn.layers.0.0, ffn.layers.1}``; ``stages.<i>.downsample.{norm, reduction}``
with the ``nn.Unfold`` channel order (c * 4 + kh * 2 + kw); ``norm<i>`` on the output levels; the relative position
index built by mmdet's ``double_step_seq``.  **Parity unpinned** (no source, no checkpoint, no fixture); the tests pin
what can be pinned: the shifted-window attention against a dense attention with an independently built window mask,
shapes and strides of the outputs.

Implementation: token GEMMs, softmax and LayerNorm are torch ops (library GEMMs on 49-token windows); the 4x4/s4
patch embedding runs on this build's conv kernel.  Eval only (no dropout / drop-path).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .bevfusion_camera import _ConvAffine
from .registry import BACKBONES


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.embed_dims, self.window_size, self.num_heads = embed_dims, window_size, num_heads
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        # mmdet's construction of the (Wh*Ww, Wh*Ww) index into the table
        seq1 = torch.arange(0, (2 * Ww - 1) * Wh, 2 * Ww - 1)
        seq2 = torch.arange(0, Ww, 1)
        rel = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
        rel_position_index = (rel + rel.T).flip(1).contiguous()
        self.register_buffer("relative_position_index", rel_position_index)
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        """x [num_windows * B, N, C]; mask [num_windows, N, N] (0 / -100) or None."""
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = attn.softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        assert 0 <= shift_size < window_size
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale)

    def window_partition(self, x):
        B, H, W, C = x.shape
        ws = self.window_size
        x = x.view(B, H // ws, ws, W // ws, ws, C)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)

    def window_reverse(self, windows, H, W):
        ws = self.window_size
        B = int(windows.shape[0] / (H * W / ws / ws))
        x = windows.view(B, H // ws, W // ws, ws, ws, -1)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)

    def forward(self, query, hw_shape):
        B, L, C = query.shape
        H, W = hw_shape
        assert L == H * W
        ws = self.window_size
        query = query.view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        query = F.pad(query, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = query.shape[1], query.shape[2]
        if self.shift_size > 0:
            shifted = torch.roll(query, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
            img_mask = torch.zeros((1, Hp, Wp, 1), device=query.device)
            slices = (slice(0, -ws), slice(-ws, -self.shift_size), slice(-self.shift_size, None))
            cnt = 0
            for h in slices:
                for w in slices:
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mask_windows = self.window_partition(img_mask).view(-1, ws * ws)
            attn_mask = mask_windows.unsqueeze(1) - mask_windows.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
        else:
            shifted, attn_mask = query, None
        windows = self.window_partition(shifted).view(-1, ws * ws, C)
        attn_windows = self.w_msa(windows, mask=attn_mask).view(-1, ws, ws, C)
        shifted = self.window_reverse(attn_windows, Hp, Wp)
        x = torch.roll(shifted, shifts=(self.shift_size, self.shift_size), dims=(1, 2)) if self.shift_size > 0 else shifted
        if pad_r > 0 or pad_b:
            x = x[:, :H, :W, :].contiguous()
        return x.view(B, H * W, C)


class _FFN(nn.Module):
    """mmcv FFN(num_fcs=2): layers = [Sequential(Linear, GELU, Dropout), Linear, Dropout]; residual outside."""

    def __init__(self, embed_dims, feedforward_channels):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Identity()),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Identity())

    def forward(self, x):
        return self.layers(x)


class SwinBlock(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = _FFN(embed_dims, feedforward_channels)

    def forward(self, x, hw_shape):
        x = x + self.attn(self.norm1(x), hw_shape)
        return x + self.ffn(self.norm2(x))


class PatchMerging(nn.Module):
    """2x2 -> 1 token: unfold (channel-major, as ``nn.Unfold``), LayerNorm(4C), Linear(4C -> 2C, no bias)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.norm = nn.LayerNorm(4 * in_channels)
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=False)

    def forward(self, x, hw_shape):
        B, L, C = x.shape
        H, W = hw_shape
        x = x.view(B, H, W, C).permute(0, 3, 1, 2)
        x = F.pad(x, (0, W % 2, 0, H % 2))                                    # "corner" padding to even sizes
        x = F.unfold(x, kernel_size=2, stride=2)                              # [B, C*4, L/4], index c*4 + kh*2 + kw
        out_hw = ((H + 1) // 2, (W + 1) // 2)
        x = self.reduction(self.norm(x.transpose(1, 2)))
        return x, out_hw


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size, qkv_bias, qk_scale, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, i % 2 == 1,
                                               qkv_bias, qk_scale) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for block in self.blocks:
            x = block(x, hw_shape)
        if self.downsample is not None:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


class _PatchEmbed(nn.Module):
    def __init__(self, in_channels, embed_dims, patch_size, patch_norm):
        super().__init__()
        self.patch_size = patch_size
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dims) if patch_norm else None
        object.__setattr__(self, "_run", _ConvAffine(self.projection, None, False))

    def forward(self, x):
        """x channels-last [B,H,W,3] -> tokens [B, H/4 * W/4, C], (H/4, W/4)."""
        ps = self.patch_size
        B, H, W, _ = x.shape
        x = F.pad(x, (0, 0, 0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
        y = self._run(x)
        hw = (y.shape[1], y.shape[2])
        y = y.reshape(B, hw[0] * hw[1], -1)
        return (self.norm(y) if self.norm is not None else y), hw


@BACKBONES.register_module
class SwinTransformer(nn.Module):
    """forward(img channels-last [B*N, H, W, 3]) -> tuple of channels-last maps [B*N, H_l, W_l, C_l] for ``out_indices``."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, **_unused):
        super().__init__()
        assert strides[0] == patch_size
        self.out_indices = tuple(out_indices)
        self.patch_embed = _PatchEmbed(in_channels, embed_dims, patch_size, patch_norm)
        self.stages = nn.ModuleList()
        c = embed_dims
        for i, (depth, heads) in enumerate(zip(depths, num_heads)):
            down = PatchMerging(c, 2 * c) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(c, heads, int(mlp_ratio * c), depth, window_size, qkv_bias, qk_scale, down))
            if down is not None:
                c = 2 * c
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    def forward(self, x):
        if self.training:
            raise RuntimeError("al3d SwinTransformer implements the eval() path only")
        x, hw_shape = self.patch_embed(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw_shape, out, out_hw = stage(x, hw_shape)
            if i in self.out_indices:
                out = getattr(self, f"norm{i}")(out)
                outs.append(out.view(-1, *out_hw, self.num_features[i]).contiguous())
        return tuple(outs)
