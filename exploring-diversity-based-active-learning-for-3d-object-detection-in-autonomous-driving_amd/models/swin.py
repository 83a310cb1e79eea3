"""Swin Transformer image backbone of BEVFusion's camera branch (BASELINE configs[4]) on
this build's token kernels (``csrc/tokens.hip``).

The reference configures ``type: SwinTransformer`` from **mmdet 2.20.0** (``mmdet/models/backbones/swin.py``;
bevfusion/configs/nuscenes/det/transfusion/secfpn/camera+lidar/swint_v0p075/default.yaml: embed_dims 96, depths
[2, 2, 6, 2], num_heads [3, 6, 12, 24], window_size 7, mlp_ratio 4, qkv_bias, patch_norm, out_indices [1, 2, 3]).
mmdet is not in the reference tree and not installed, so this implements the PUBLISHED algorithm (Liu et al., "Swin
Transformer", ICCV 2021) with mmdet's module tree and conventions where they are observable from its checkpoints'
key names: ``patch_embed.projection`` / ``.norm``; ``stages.<i>.blocks.<j>.{norm1, attn.w_msa.{qkv, proj,
relative_position_bias_table}, norm2, ffn.layers.0.0, ffn.layers.1}``; ``stages.<i>.downsample.{norm, reduction}``
with the ``nn.Unfold`` channel order (c * 4 + kh * 2 + kw); ``norm<i>`` on the output levels; the relative position
index built by mmdet's ``double_step_seq``.  **Parity unpinned** (no source, no checkpoint, no fixture).

How a block runs (seven launches -- five at embed dim 96, where LN2 / fc1 / fc2 are ONE kernel, al3d_tok_mlp_f16x3 --; no torch
op touches an activation):
  LN1 gathered into (shifted) window order, padding rows zero, pair rows   al3d_tok_layernorm_f32
  qkv projection                                                           al3d_tok_linear_f16x3
  49 x 49 attention per (window, head) with bias + region mask, pair rows  al3d_tok_window_attention_f32
  output projection + residual, scattered back to token order              al3d_tok_linear_f16x3
  LN2, pair rows                                                           al3d_tok_layernorm_f32
  fc1 + exact GELU, pair rows                                              al3d_tok_linear_f16x3
  fc2 + residual                                                           al3d_tok_linear_f16x3
Patch merging = LN over the gathered 2 x 2 neighbourhood + one GEMM (weights re-ordered once from mmdet's
channel-major unfold order to piece-major); the patch embedding is a gather of 4 x 4 patches into rows of 48
(al3d_tok_patch_rows_f32) + one GEMM.  The modules only hold parameters (state-dict compatible); ``tests/swin_torch.py`` is the torch restatement the
kernels are checked against.  Eval only (no dropout / drop-path).
"""
import torch
from torch import nn

from .. import token_ops as T
from .bevfusion_camera import _versions
from .registry import BACKBONES


import os as _os
SWIN_STREAMS = int(_os.environ.get("AL3D_SWIN_STREAMS", "1"))
_STREAMS = {}


def _stream_pool(device, i):
    key = (torch.device(device).index or 0, i)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=device)
    return _STREAMS[key]


class _Packed:
    """Per-module cache of kernel-format weights, rebuilt when the device or a parameter changes."""

    def __init__(self):
        self._key, self._val = None, None

    def get(self, device, mods, build):
        key = (device, _versions(*mods))
        if self._key != key:
            self._val, self._key = build(), key
        return self._val


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.embed_dims, self.window_size, self.num_heads = embed_dims, window_size, num_heads
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        # mmdet's construction of the (Wh*Ww, Wh*Ww) index into the table (kept for state dicts and the tests; the
        # attention kernel evaluates the same index arithmetically)
        seq1 = torch.arange(0, (2 * Ww - 1) * Wh, 2 * Ww - 1)
        seq2 = torch.arange(0, Ww, 1)
        rel = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
        rel_position_index = (rel + rel.T).flip(1).contiguous()
        self.register_buffer("relative_position_index", rel_position_index)
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        object.__setattr__(self, "_pk", _Packed())

    def packed(self, device):
        return self._pk.get(device, (self, self.qkv, self.proj), lambda: (
            T.PackedLinear(self.qkv.weight, self.qkv.bias), T.PackedLinear(self.proj.weight, self.proj.bias),
            self.relative_position_bias_table.detach().float().contiguous()))

    def qkv_bias_rows(self, device):
        """The qkv bias as the q / k / v row of a padded window position (zeros for a layer without bias)."""
        if getattr(self, "_pkb", None) is None:
            object.__setattr__(self, "_pkb", _Packed())
        return self._pkb.get(device, (self.qkv,), lambda: (
            torch.zeros(3 * self.embed_dims, device=device) if self.qkv.bias is None
            else self.qkv.bias.detach().float().to(device).contiguous()))

    def packed_fused(self, device, norm):
        """LN1 + qkv + attention + proj in the fused kernel's format (``al3d_tok_attn_block_f16x3``; embed dim 96 / 192)."""
        if getattr(self, "_pkf", None) is None:
            object.__setattr__(self, "_pkf", _Packed())
        return self._pkf.get(device, (self, self.qkv, self.proj, norm), lambda: T.PackedAttnBlock(
            norm.weight.to(device), norm.bias.to(device), norm.eps, self.qkv.weight.to(device), self.qkv.bias,
            self.proj.weight.to(device), self.proj.bias, self.relative_position_bias_table.to(device)))


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        assert 0 <= shift_size < window_size
        if window_size != 7 or embed_dims != 32 * num_heads:
            raise NotImplementedError("the attention kernel is built for 7 x 7 windows and 32-channel heads (Swin-T/S/B)")
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale)


# AL3D_SWIN_ATTN=split: LN1 / qkv / attention / proj as four launches everywhere (default "fused": one kernel at the
# embed dims it is built for -- stages 0-1, whose four launches are bound by the activation bytes they move)
FUSED_ATTN = _os.environ.get("AL3D_SWIN_ATTN", "fused") != "split"
FUSED_ATTN_DIMS = tuple(int(v) for v in _os.environ.get("AL3D_SWIN_ATTN_DIMS", "96,192").split(",") if v)
# AL3D_SWIN_PATCH=split: patch rows + token GEMM + LayerNorm as three launches (default "fused": one kernel at embed dim 96)
FUSED_PATCH_EMBED = _os.environ.get("AL3D_SWIN_PATCH", "fused") != "split"
# AL3D_SWIN_ROWS=window: the unfused attention half on window-ordered, padded rows (LN1 gathers through the row map, the
# projection scatters back); default "token": its GEMMs run on the map's tokens and the attention kernel does the gathering
TOKEN_ORDER = _os.environ.get("AL3D_SWIN_ROWS", "token") != "window"
# AL3D_SWIN_MLP=split: LN2 / fc1 / fc2 as three launches everywhere (default "fused": one kernel where it is faster)
FUSED_MLP = _os.environ.get("AL3D_SWIN_MLP", "fused") != "split"
# embed dims 96 (32-token waves) and 192 (16-token waves on the 16 x 16 x 32 product); AL3D_SWIN_MLP_DIMS=96 keeps stage 1 split
FUSED_MLP_DIMS = tuple(int(v) for v in _os.environ.get("AL3D_SWIN_MLP_DIMS", "96,192").split(",") if v)


class _FFN(nn.Module):
    """mmcv FFN(num_fcs=2): layers = [Sequential(Linear, GELU, Dropout), Linear, Dropout]; residual outside."""

    def __init__(self, embed_dims, feedforward_channels):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Identity()),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Identity())
        object.__setattr__(self, "_pk", _Packed())

    def packed(self, device):
        fc1, fc2 = self.layers[0][0], self.layers[1]
        return self._pk.get(device, (fc1, fc2), lambda: (T.PackedLinear(fc1.weight, fc1.bias),
                                                         T.PackedLinear(fc2.weight, fc2.bias)))

    def packed_fused(self, device, norm):
        """LN2 + fc1 + fc2 in the fused kernel's format (``al3d_tok_mlp_f16x3``; embed dims 96 / 192)."""
        fc1, fc2 = self.layers[0][0], self.layers[1]
        if getattr(self, "_pkf", None) is None:
            object.__setattr__(self, "_pkf", _Packed())
        return self._pkf.get(device, (fc1, fc2, norm), lambda: T.PackedMlp(
            norm.weight.to(device), norm.bias.to(device), norm.eps, fc1.weight.to(device), fc1.bias, fc2.weight.to(device), fc2.bias))


class SwinBlock(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = _FFN(embed_dims, feedforward_channels)

    def forward(self, x, geom):
        """x [B * H * W, C] f32 token rows, updated IN PLACE; geom = _Geometry of the stage."""
        msa = self.attn.w_msa
        n1, n2 = self.norm1, self.norm2
        if FUSED_ATTN and x.shape[-1] in FUSED_ATTN_DIMS:
            # LN1 + qkv + attention + proj + residual as one kernel: q, k, v and the attention output stay on the CU
            T.attn_block(x, geom.B, geom.H, geom.W, msa.packed_fused(x.device, n1), self.attn.shift_size, msa.scale)
        elif TOKEN_ORDER:
            # LN1, qkv and proj on the map's own tokens; the attention kernel gathers its windows (shift, padding: a padded
            # position's q / k / v is the qkv bias) and writes token order back -- no GEMM row for the window padding
            qkv_w, proj_w, table = msa.packed(x.device)
            xw = T.layernorm(x, n1.weight, n1.bias, n1.eps, pair=True)
            qkv = T.linear(xw, qkv_w, a_pair=True)
            ao = T.window_attention_tokens(qkv, msa.qkv_bias_rows(x.device), table, geom.B, geom.H, geom.W, msa.num_heads,
                                           self.attn.shift_size, msa.scale, pair=True)
            T.linear(ao, proj_w, a_pair=True, residual=x, out=x)
        else:
            rowmap, (nwy, nwx) = geom.window_map(self.attn.shift_size)
            qkv_w, proj_w, table = msa.packed(x.device)
            xw = T.layernorm(x, n1.weight, n1.bias, n1.eps, rowmap=rowmap, zero_out=True, pair=True)
            qkv = T.linear(xw, qkv_w, a_pair=True)
            ao = T.window_attention(qkv, table, msa.num_heads, nwy, nwx, self.attn.shift_size, msa.scale, pair=True)
            T.linear(ao, proj_w, a_pair=True, residual=x, rowmap=rowmap, out=x)
        if FUSED_MLP and x.shape[-1] in FUSED_MLP_DIMS:
            # stages 0-1: LN2 + fc1 + GELU + fc2 + residual as one kernel (the hidden activation stays in registers)
            T.mlp(x, self.ffn.packed_fused(x.device, n2))
            return x
        fc1_w, fc2_w = self.ffn.packed(x.device)
        xn = T.layernorm(x, n2.weight, n2.bias, n2.eps, pair=True)
        hid = T.linear(xn, fc1_w, a_pair=True, act="gelu", out_pair=True)
        T.linear(hid, fc2_w, a_pair=True, residual=x, out=x)
        return x


class PatchMerging(nn.Module):
    """2x2 -> 1 token: unfold (channel-major, as ``nn.Unfold``), LayerNorm(4C), Linear(4C -> 2C, no bias)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = nn.LayerNorm(4 * in_channels)
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=False)
        object.__setattr__(self, "_pk", _Packed())

    def packed(self, device):
        C = self.in_channels

        def build():
            # the gather concatenates whole tokens (piece-major, index j * C + c); mmdet's unfold order is c * 4 + j
            w = self.reduction.weight.detach().float().view(-1, C, 4).permute(0, 2, 1).reshape(-1, 4 * C)
            g = self.norm.weight.detach().float().view(C, 4).t().reshape(-1).contiguous()
            b = self.norm.bias.detach().float().view(C, 4).t().reshape(-1).contiguous()
            return T.PackedLinear(w, None), g, b
        return self._pk.get(device, (self.norm, self.reduction), build)

    def forward(self, x, geom):
        rowmap, out_hw = geom.merge_map()
        w, g, b = self.packed(x.device)
        xm = T.layernorm(x, g, b, self.norm.eps, rowmap=rowmap, G=4, pair=True)
        return T.linear(xm, w, a_pair=True), out_hw


class _Geometry:
    """Row maps of one stage's token grid (B maps of H x W), built on the host once per shape and kept on the device."""
    _cache = {}

    def __init__(self, B, H, W, ws, device):
        self.B, self.H, self.W, self.ws, self.device = B, H, W, ws, device

    @classmethod
    def of(cls, B, H, W, ws, device):
        key = (B, H, W, ws, str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(B, H, W, ws, device)
            cls._cache[key]._maps = {}
        return cls._cache[key]

    def window_map(self, shift):
        k = ("win", shift)
        if k not in self._maps:
            m, grid = T.window_rowmap(self.B, self.H, self.W, self.ws, shift)
            self._maps[k] = (torch.from_numpy(m).to(self.device), grid)
        return self._maps[k]

    def merge_map(self):
        if "merge" not in self._maps:
            m, hw = T.merge_rowmap(self.B, self.H, self.W)
            self._maps["merge"] = (torch.from_numpy(m).to(self.device), hw)
        return self._maps["merge"]


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size, qkv_bias, qk_scale, downsample):
        super().__init__()
        self.window_size = window_size
        self.blocks = nn.ModuleList([SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, i % 2 == 1,
                                               qkv_bias, qk_scale) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, B, hw_shape):
        geom = _Geometry.of(B, hw_shape[0], hw_shape[1], self.window_size, x.device)
        for block in self.blocks:
            x = block(x, geom)
        if self.downsample is not None:
            x_down, down_hw = self.downsample(x, geom)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


class _PatchEmbed(nn.Module):
    """mmdet ``PatchEmbed``: Conv2d(in, embed, 4, stride 4) (+ LayerNorm) -- here a gather of 4 x 4 patches into rows of
    48 and one token GEMM."""

    def __init__(self, in_channels, embed_dims, patch_size, patch_norm):
        super().__init__()
        if in_channels != 3 or patch_size != 4:
            raise NotImplementedError("the patch-row kernel is built for 3-channel images and 4 x 4 patches")
        self.patch_size = patch_size
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dims) if patch_norm else None
        object.__setattr__(self, "_pk", _Packed())

    def forward(self, x):
        """x channels-last [B,H,W,3] -> token rows [B * H/4 * W/4, C], (H/4, W/4)."""
        pr = self.projection
        if FUSED_PATCH_EMBED and pr.out_channels == 96 and x.shape[2] % 4 == 0:
            # projection + LayerNorm as one kernel: the image is read once, the tokens are written once
            if getattr(self, "_pkf", None) is None:
                object.__setattr__(self, "_pkf", _Packed())
            n = self.norm
            pk = self._pkf.get(x.device, (pr,) + ((n,) if n is not None else ()), lambda: T.PackedPatchEmbed(
                pr.weight.to(x.device), pr.bias, None if n is None else n.weight, None if n is None else n.bias,
                1e-5 if n is None else n.eps))
            return T.patch_embed(x, pk)
        w = self._pk.get(x.device, (pr,), lambda: T.PackedLinear(
            pr.weight.detach().permute(0, 2, 3, 1).reshape(pr.out_channels, -1), pr.bias))      # [C, (ky, kx, c)]
        rows, hw = T.patch_rows(x, pair=True)
        y = T.linear(rows, w, a_pair=True)
        if self.norm is not None:
            y = T.layernorm(y, self.norm.weight, self.norm.bias, self.norm.eps)
        return y, hw


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
        n = SWIN_STREAMS
        if n > 1 and x.is_cuda and x.shape[0] >= 2 * n:
            # Images are independent: run the batch as n chunks on n streams.  The stage-0 / stage-1 GEMMs spend ~40 % of
            # their time in a store tail that keeps a workgroup's LDS and registers but no pipe busy; with a
            # second launch queue the other chunk's kernels run in those holes.
            main = torch.cuda.current_stream(x.device)
            chunks = torch.chunk(x, n, dim=0)
            outs = []
            for i, c in enumerate(chunks):
                st = _stream_pool(x.device, i)
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs.append(self._forward_one(c.contiguous()))
            for i, o in enumerate(outs):
                main.wait_stream(_stream_pool(x.device, i))
                for t in o:
                    t.record_stream(main)
            return tuple(torch.cat([o[l] for o in outs], dim=0) for l in range(len(outs[0])))
        return self._forward_one(x)

    def _forward_one(self, x):
        B = x.shape[0]
        x, hw_shape = self.patch_embed(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw_shape, out, out_hw = stage(x, B, hw_shape)
            if i in self.out_indices:
                n = getattr(self, f"norm{i}")
                out = T.layernorm(out, n.weight, n.bias, n.eps)
                outs.append(out.view(B, *out_hw, self.num_features[i]))
        return tuple(outs)
