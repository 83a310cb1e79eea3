"""Device-side detector primitives (torch tensors in, HIP kernels underneath).

Activations are NHWC float32.  Weight packing helpers turn reference-layout
parameters (``Conv2d.weight [Cout,Cin,k,k]``, ``ConvTranspose2d.weight [Cin,Cout,2,2]``,
eval ``BatchNorm2d``) into the kernel layouts once, outside the hot loop.
"""
import torch

from . import lib
from .selector_ops import _dev, _ptr, _stream


# ------------------------------------------------------------------ packing (one-off)
def pack_conv_weight(w):
    """[Cout,Cin,k,k] -> [Cout,k*k,Cin] contiguous."""
    co, ci, kh, kw = w.shape
    return w.detach().permute(0, 2, 3, 1).reshape(co, kh * kw, ci).contiguous().float()


def pack_deconv_weight(w):
    """ConvTranspose2d [Cin,Cout,2,2] -> [Cout,4,Cin] with tap = dy*2+dx."""
    ci, co, kh, kw = w.shape
    assert kh == 2 and kw == 2
    return w.detach().permute(1, 2, 3, 0).reshape(co, 4, ci).contiguous().float()


def fold_bn(bn):
    """eval BatchNorm -> (scale, shift): y = x*scale + shift."""
    inv = torch.rsqrt(bn.running_var.detach().double() + bn.eps)
    g = bn.weight.detach().double() if bn.weight is not None else torch.ones_like(inv)
    b = bn.bias.detach().double() if bn.bias is not None else torch.zeros_like(inv)
    scale = g * inv
    shift = b - bn.running_mean.detach().double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


def split_bf16x3(w_packed):
    """f32 packed weights (device) -> bf16 [3, *shape] exact three-way split (hi, mid, lo)."""
    w = _dev(w_packed, torch.float32, "w")
    out = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    lib.call("al3d_split_bf16x3", _ptr(w), w.numel(), _ptr(out), _stream())
    return out


def split_f16x3(w_packed, scale=None):
    """f32 packed weights (device) -> (f16 [2, *shape] planes (wh, wl), scale * 2^-s).

    ``s`` is the power of two that brings max|w| just under 2^14 (f16's exponent range is spent
    on the weights once, here, so the kernel never depends on it); it is folded back into the
    per-channel scale the kernel multiplies its accumulator with -- exact."""
    import math
    w = _dev(w_packed, torch.float32, "w")
    amax = float(w.abs().max()) if w.numel() else 0.0
    if not math.isfinite(amax):
        raise lib.Al3dError("split_f16x3: non-finite weight")
    s = 14 - math.frexp(amax)[1] if amax > 0.0 else 0          # amax <= 2^e  ->  amax * 2^s <= 2^14
    s = max(-100, min(100, s))
    out = torch.empty((2,) + tuple(w.shape), dtype=torch.float16, device=w.device)
    lib.call("al3d_split_f16x3", _ptr(w), w.numel(), s, _ptr(out), _stream())
    cout = w.shape[0]
    base = torch.ones(cout, dtype=torch.float32, device=w.device) if scale is None else \
        _dev(scale, torch.float32, "scale")
    return out, (base.double() * 2.0 ** (-s)).float().contiguous()


# Arithmetic of the MFMA conv kernels (AL3D_MATH):
#   "f16x3"  (default) three f16 products per MAC (fp32-class, activations < 65504) in the dense
#            neck + head and in the sparse encoder
#   "bf16x6" fp32-faithful six-product split on the bf16 matrix cores everywhere (full fp32 range)
#   "f32"    fp32-input MFMA (bitwise an fp32 FMA chain)
import os as _os
MATH = _os.environ.get("AL3D_MATH", "f16x3")
if MATH not in ("f16x3", "bf16x6", "f32"):
    raise lib.Al3dError(f"AL3D_MATH={MATH!r}: expected f16x3, bf16x6 or f32")


def sparse_math():
    """Arithmetic of the sparse encoder under the current MATH."""
    return MATH


# structure of the f16x3 dense kernels: "auto" = 3x3/s1 layers on the fragment-streamed halo kernel, every other
# geometry (stride-2 entry, 1x1 / deconv deblocks, fused head) on the LDS-DMA kernel; "stream" = round 1's policy
# (generic launches with >= 24 steps stream their weights in fragment order, the others stage both tiles through
# LDS), "bstream" = streamed weights everywhere, "frag" = only the 3x3 layers, "lds" = the LDS-staged kernels
# everywhere.  The same bits in all of them; kept for A/B
DENSE = _os.environ.get("AL3D_DENSE", "auto")


class F16x3Packed:
    """f16x3 weights in MFMA fragment order for the kernels that stream them from L2.
    kind "frag3x3": [2,Cout/32,Cin/16,9,64,8] (3x3/s1/p1); kind "bstream": [2,ceil(Cout/128)*4,taps,
    Cin/16,64,8] (any other geometry, zero rows beyond Cout)."""
    dtype = torch.float16

    def __init__(self, kind, data, cout, taps, cin):
        self.kind, self.data, self.cout, self.taps, self.cin = kind, data, cout, taps, cin


def pack_frag_f16x3(planes):
    """f16 planes [2,Cout,9,Cin] -> fragment order for the 3x3/s1/p1 kernel (every wave streams its B
    operands from L2, no LDS staging of weights)."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, taps, cin = planes.shape
    assert taps == 9
    out = torch.empty((2, cout // 32, cin // 16, 9, 64, 8), dtype=torch.float16, device=planes.device)
    lib.call("al3d_pack_f16x3_frag", _ptr(planes), cout, cin, _ptr(out), _stream())
    return F16x3Packed("frag3x3", out, cout, 9, cin)


def pack_frag16_f16x3(planes):
    """f16 planes [2,Cout,9,Cin] -> 16x16x32 fragment order for the 3x3/s1/p1 kernel on the narrow MFMA shape."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, taps, cin = planes.shape
    assert taps == 9
    out = torch.empty((2, cout // 16, cin // 32, 9, 64, 8), dtype=torch.float16, device=planes.device)
    lib.call("al3d_pack_f16x3_frag16", _ptr(planes), cout, cin, _ptr(out), _stream())
    return F16x3Packed("frag16", out, cout, 9, cin)


def pack_bstream_f16x3(planes):
    """f16 planes [2,Cout,taps,Cin] -> fragment order for the streamed-weight kernel of the other geometries."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, taps, cin = planes.shape
    n = lib.load().al3d_pack_f16x3_bstream_elems(cout, taps, cin)
    if n <= 0:
        raise lib.Al3dError(f"pack_bstream_f16x3: unsupported shape Cout={cout} taps={taps} Cin={cin}")
    out = torch.empty((2, (cout + 127) // 128 * 4, taps, cin // 16, 64, 8), dtype=torch.float16,
                      device=planes.device)
    assert out.numel() == n
    lib.call("al3d_pack_f16x3_bstream", _ptr(planes), cout, taps, cin, _ptr(out), _stream())
    return F16x3Packed("bstream", out, cout, taps, cin)


def pack_dma_f16x3(planes):
    """f16 planes [2,Cout,taps,Cin] -> per-step LDS images for the LDS-DMA kernel of the generic geometries."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, taps, cin = planes.shape
    n = lib.load().al3d_pack_f16x3_bstream_elems(cout, taps, cin)
    if n <= 0:
        raise lib.Al3dError(f"pack_dma_f16x3: unsupported shape Cout={cout} taps={taps} Cin={cin}")
    out = torch.empty(((cout + 127) // 128, taps, cin // 16, 2, 128, 16), dtype=torch.float16, device=planes.device)
    assert out.numel() == n
    lib.call("al3d_pack_f16x3_dma", _ptr(planes), cout, taps, cin, _ptr(out), _stream())
    return F16x3Packed("dma", out, cout, taps, cin)


def pack_wino_f16x3(w_packed, scale=None):
    """f32 packed 3x3 weights [Cout, 9, Cin] -> Winograd F(2x2, 3x3) weights U = G g G^T (float64 on the host side of the
    split, then the usual f16x3 planes) in fragment order for ``al3d_conv3x3_nhwc_f16x3_wino``, + the scale to hand it."""
    w = _dev(w_packed, torch.float32, "w")
    cout, taps, cin = w.shape
    assert taps == 9
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ak,okqc,bq->oabc", G, w.double().view(cout, 3, 3, cin), G).reshape(cout, 16, cin).float().contiguous()
    planes, scale = split_f16x3(U, scale)
    out = torch.empty((2, cout // 32, cin // 16, 16, 64, 8), dtype=torch.float16, device=w.device)
    lib.call("al3d_pack_f16x3_wino", _ptr(planes), cout, cin, _ptr(out), _stream())
    return F16x3Packed("wino", out, cout, 9, cin), scale


def wino_ok(cout, cin, ksize, stride, pad):
    return (ksize, stride, pad) == (3, 1, 1) and cout % 64 == 0 and cin % 16 == 0


class GldsPacked:
    """f16x3 sparse-conv weights in the LDS image order of the DMA-gather kernel
    (al3d_sp_pack_glds_f16x3: [K][Cin/16][2][ceil32(Cout)][16] f16, halves swizzled)."""
    dtype = torch.float16

    def __init__(self, data, cout, K, cin):
        self.data, self.cout, self.K, self.cin = data, cout, K, cin


def pack_glds_f16x3(planes):
    """f16 planes [2,Cout,K,Cin] (split_f16x3 of the [Cout,K,Cin] weights) -> GldsPacked."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, K, cin = planes.shape
    n = lib.load().al3d_sp_pack_glds_f16x3_elems(cout, K, cin)
    if n <= 0:
        raise lib.Al3dError(f"pack_glds_f16x3: unsupported shape Cout={cout} K={K} Cin={cin}")
    out = torch.empty((n,), dtype=torch.float16, device=planes.device)
    lib.call("al3d_sp_pack_glds_f16x3", _ptr(planes), cout, K, cin, _ptr(out), _stream())
    return GldsPacked(out, cout, K, cin)


def frag_ok(cout, cin, ksize, stride, pad):
    return ksize == 3 and stride == 1 and pad == 1 and cout % 128 == 0 and cin % 32 == 0


def pack_dense(w_packed, scale=None, ksize=None, stride=None, pad=None):
    """Packed f32 weights + folded-BN scale -> (weights in the dense kernels' format for MATH,
    the scale to hand them).  With the layer geometry given (``ksize="deconv"`` for the 2x2
    transposed conv), f16x3 weights are put in fragment order for the streamed-weight kernels."""
    if MATH == "f16x3":
        if DENSE == "wino" and ksize not in (None, "deconv") and wino_ok(w_packed.shape[0], w_packed.shape[2], ksize, stride, pad):
            return pack_wino_f16x3(w_packed, scale)
        planes, scale = split_f16x3(w_packed, scale)
        if ksize is not None and DENSE != "lds":
            if ksize != "deconv" and frag_ok(planes.shape[1], planes.shape[3], ksize, stride, pad):
                if DENSE == "frag16" and planes.shape[3] % 64 == 0:
                    return pack_frag16_f16x3(planes), scale
                return pack_frag_f16x3(planes), scale
            # streamed weights pay off once a launch has enough steps to amortise the deeper prologue:
            # stride-2 3x3 (72 steps) -12 %, fused head (32) -5 %, 1x1 deblock (8) +8 % -> LDS-staged
            if DENSE in ("auto", "dma", "wino"):
                return pack_dma_f16x3(planes), scale
            steps = planes.shape[2] * planes.shape[3] // 16
            if DENSE == "bstream" or (DENSE == "stream" and ksize != "deconv" and steps >= 24):
                return pack_bstream_f16x3(planes), scale
        return planes, scale
    if MATH == "bf16x6":
        return split_bf16x3(w_packed), scale
    return w_packed, scale


# sparse-conv structure.  f16x3: "auto" = per channel pair whichever kernel measured fastest on the real rulebooks
# (RNG_PAIRS: range-gather LDS-DMA kernel for the 27-tap submanifold layers; GLDS_PAIRS: per-tap LDS-DMA gather; else
# the register-gather wave kernel), "glds" / "wave2" = one of those everywhere, "rng" = range gather wherever built
# (same bits in all of them for Cin <= 32; see al3d_sp_conv_rng_f16x3 for Cin = 64).  bf16x6: "auto" = the software-pipelined wave kernel; "wave" (unpipelined
# wave kernel) and "tile" (LDS-staged 128-row tile) give the same bits, for A/B
SPCONV = _os.environ.get("AL3D_SPCONV", "auto")
GLDS_PAIRS = {(32, 32), (64, 64)}
if _os.environ.get("AL3D_GLDS_PAIRS"):           # dev override, e.g. "32x32,64x64,128x128"
    GLDS_PAIRS = {tuple(int(v) for v in t.split("x")) for t in _os.environ["AL3D_GLDS_PAIRS"].split(",")}


# row format of the f16x3 sparse encoder's activations between layers (csrc/sp_rows.h): "pair" = the two f16 planes
# of the arithmetic, split once in the producer's epilogue; "f32" = plain rows, split per gathered (row, tap)
SPROWS = _os.environ.get("AL3D_SPROWS", "pair")
# the same format for the dense neck's maps where the consumer is the LDS-DMA kernel (block outputs -> stride-2 conv /
# deblocks, concat map -> fused head): "pair" | "f32"
DPIX = _os.environ.get("AL3D_DPIX", "pair")
IO_IN_PAIR, IO_OUT_PAIR, IO_RES_PAIR = 1, 2, 4


def rows_convert(x, to_pair):
    """[n, C] f32 rows <-> pair rows (same shape and dtype; C % 8 == 0)."""
    x = _dev(x, torch.float32, "x")
    out = torch.empty_like(x)
    lib.call("al3d_sp_rows_convert_f16x3", _ptr(x), x.shape[0], x.shape[1], 1 if to_pair else 0, _ptr(out), _stream())
    return out


# the range-gather form of the LDS-DMA kernel (27-tap submanifold layers of these channel pairs; "rng" = wherever it
# is built, "auto" = RNG_PAIRS, measured)
RNG_BUILT = {(32, 32), (64, 64)}
RNG_PAIRS = {(32, 32)}        # measured: ahead of the per-tap kernel at 32 channels, behind at 64
if _os.environ.get("AL3D_RNG_PAIRS") is not None:   # dev override, e.g. "32x32,64x64" or "" for none
    RNG_PAIRS = {tuple(int(v) for v in t.split("x")) for t in _os.environ["AL3D_RNG_PAIRS"].split(",") if t}


def sparse_rng(cin, cout):
    """True when the 27-tap submanifold f16x3 layer cin -> cout runs on the range-gather kernel."""
    if MATH != "f16x3" or SPCONV not in ("auto", "rng"):
        return False
    return (cin, cout) in (RNG_BUILT if SPCONV == "rng" else RNG_PAIRS & RNG_BUILT)


# Block-staged kernel (csrc/spconv_blk.hip): 27-tap submanifold layers of these channel pairs stage the union of a
# 128-row chunk's neighbourhoods once; the encoder then numbers the level's rows column by column
# (al3d_sp_down_sites_blocked).  Built, bit-identical, and MEASURED SLOWER than the range / per-tap kernels on lidar
# data (round 5, DESIGN 5.3: fragment reads through arbitrary local indices are LDS-bank-conflict bound, and column order
# makes 91 % instead of 72 % of the (tile, tap) pairs live), so no pair takes it by default.
# AL3D_BLK_PAIRS: opt-in, e.g. "32x32,64x64,128x128"
BLK_BUILT = {(32, 32), (64, 64), (128, 128)}
BLK_PAIRS = set()
if _os.environ.get("AL3D_BLK_PAIRS") is not None:
    BLK_PAIRS = {tuple(int(v) for v in t.split("x")) for t in _os.environ["AL3D_BLK_PAIRS"].split(",") if t}


# Tap-mask row order (al3d_sp_mask_window_sort): levels whose 27-tap submanifold layers have these output widths get their rows
# re-numbered by neighbour mask inside windows of MASK_SORT_WINDOW raster rows, so that the rows of a 32-row tile lack the same
# taps (the per-tap gather kernels of the 64- and 128-channel levels execute 10-15 % fewer (tile, tap) pairs: 64 -> 64 layers
# -11 %, 128 -> 128 -10 %, bench +2.3 % at windows of 16384; level 1's range-gather kernel needs raster order: +9 % slower
# there).  AL3D_MASK_SORT: e.g. "64,128" / "" for none; AL3D_MASK_SORT_WINDOW: 1024 | 4096 | 8192 | 16384
MASK_SORT = {64, 128}
if _os.environ.get("AL3D_MASK_SORT") is not None:
    MASK_SORT = {int(v) for v in _os.environ["AL3D_MASK_SORT"].split(",") if v}
MASK_SORT_WINDOW = int(_os.environ.get("AL3D_MASK_SORT_WINDOW", "16384"))
MASK_SORT_WINDOWS = {}         # per width, e.g. AL3D_MASK_SORT_WINDOWS="64:4096,128:16384" (else MASK_SORT_WINDOW for all)
if _os.environ.get("AL3D_MASK_SORT_WINDOWS"):
    MASK_SORT_WINDOWS = {int(a): int(b) for a, b in (t.split(":") for t in _os.environ["AL3D_MASK_SORT_WINDOWS"].split(",") if t)}


BLK_ORDER_ONLY = _os.environ.get("AL3D_BLK_ORDER_ONLY", "0") == "1"   # dev: column order for these pairs' levels, old kernels


def sparse_blk_order(cin, cout, K=27):
    """True when the level whose 27-tap submanifold layers are cin -> cout is numbered column by column."""
    return MATH == "f16x3" and SPCONV == "auto" and K == 27 and (cin, cout) in (BLK_PAIRS & BLK_BUILT)


def sparse_blk(cin, cout, K=27):
    """True when the 27-tap submanifold f16x3 layer cin -> cout runs on the block-staged kernel."""
    return sparse_blk_order(cin, cout, K) and not BLK_ORDER_ONLY


class BlkPlan:
    """Per-level plan of the block-staged kernel (al3d_sp_block_plan): chunk headers, staged row lists, local indices."""

    def __init__(self, hdr, rows, loc, R, cap):
        self.hdr, self.rows, self.loc, self.R, self.cap = hdr, rows, loc, R, cap


def block_shape(cin, cout):
    import ctypes
    R, cap = ctypes.c_int(0), ctypes.c_int(0)
    lib.call("al3d_sp_block_shape", cin, cout, ctypes.byref(R), ctypes.byref(cap))
    return R.value, cap.value


def block_plan(nbr, n_out, cin, cout):
    """Plan of a tiled 27-tap table for the block-staged kernel of cin -> cout."""
    dev = nbr.device
    R, cap = block_shape(cin, cout)
    chunks = (max(n_out, 1) + R - 1) // R
    hdr = torch.empty((chunks, 2), dtype=torch.int32, device=dev)
    rows = torch.empty((chunks, cap), dtype=torch.int32, device=dev)
    loc = torch.empty((chunks, 27, R), dtype=torch.int16, device=dev)
    lib.call("al3d_sp_block_plan", _ptr(nbr), nbr.shape[1], 27, n_out, R, cap, _ptr(hdr), _ptr(rows), _ptr(loc), _stream())
    return BlkPlan(hdr, rows, loc, R, cap)


# Level-0 layers (16 input channels, 27 taps) on raster-ordered rows (csrc/spconv_l0.hip): AL3D_L0 = "raster"
# (default: the encoder renumbers the voxelizer's rows in raster order and runs these layers as item streams with
# LDS-resident weights) | "off" (first-appearance order, register-gather kernel: round 3's path).  R16_COUTS: the output
# widths that take the item-stream kernel (16: the five submanifold layers; 32: the strided 16 -> 32 layer)
L0 = _os.environ.get("AL3D_L0", "raster")
if L0 not in ("raster", "off"):
    raise lib.Al3dError(f"AL3D_L0={L0!r}: expected raster or off")
R16_COUTS = {16}
if _os.environ.get("AL3D_R16_COUTS") is not None:    # dev override, e.g. "16,32" or "" for none
    R16_COUTS = {int(v) for v in _os.environ["AL3D_R16_COUTS"].split(",") if v}
R16_TPW = int(_os.environ.get("AL3D_R16_TPW", "0"))   # tiles per wave (0: the library's default)
# row format between two item-stream layers of level 0: "f32" (default: the encoder's output keeps the bits of every other
# kernel structure and of round 3) | "pair" (the split runs once, in the producer: -9 % per layer at 32 frames per launch,
# nothing measurable end to end, and the stored activations lose 1-2 bits: embedding moves by <= 9.5e-7 of its scale)
L0_ROWS = _os.environ.get("AL3D_L0_ROWS", "f32")


def sparse_raster():
    """True when the encoder renumbers its level-0 rows in raster order."""
    return MATH == "f16x3" and SPCONV == "auto" and L0 == "raster"


def sparse_r16(cin, cout, K=27):
    """True when the 27-tap f16x3 layer 16 -> cout runs on the item-stream kernel."""
    return sparse_raster() and cin == 16 and K == 27 and cout in R16_COUTS


class R16Packed:
    """f16x3 weights of a 16-input-channel 27-tap layer as the item-stream kernel's LDS image
    (al3d_sp_pack_r16_f16x3: [27][2 planes][2 k-halves][Cout][8] f16)."""
    dtype = torch.float16

    def __init__(self, data, cout):
        self.data, self.cout, self.K, self.cin = data, cout, 27, 16


def pack_r16_f16x3(planes):
    """f16 planes [2,Cout,27,16] (split_f16x3 of the [Cout,27,16] weights) -> R16Packed."""
    planes = _dev(planes, torch.float16, "planes")
    _, cout, K, cin = planes.shape
    n = lib.load().al3d_sp_pack_r16_f16x3_elems(cout)
    if n <= 0 or K != 27 or cin != 16:
        raise lib.Al3dError(f"pack_r16_f16x3: unsupported shape Cout={cout} K={K} Cin={cin}")
    out = torch.empty((n,), dtype=torch.float16, device=planes.device)
    lib.call("al3d_sp_pack_r16_f16x3", _ptr(planes), cout, _ptr(out), _stream())
    return R16Packed(out, cout)


def raster_perm(coords, batch, shape, frame_rows_max=0):
    """coords [n,4] i32 (b,z,y,x) -> (perm [n] i32: raster position -> row, coords in raster order).
    frame_rows_max > 0: the rows are frame-sorted with at most that many rows per frame (the voxelizer's output and its
    voxel cap): the sort then runs in LDS, one workgroup per frame."""
    coords = _dev(coords, torch.int32, "coords")
    n = coords.shape[0]
    D_, H_, W_ = [int(v) for v in shape]
    perm = torch.empty((n,), dtype=torch.int32, device=coords.device)
    out = torch.empty_like(coords)
    ws = torch.empty(max(int(lib.load().al3d_sp_raster_perm_workspace_bytes(n, batch, D_, H_)), 1), dtype=torch.uint8,
                     device=coords.device)
    lib.call("al3d_sp_raster_perm", _ptr(coords), n, batch, D_, H_, W_, int(frame_rows_max), _ptr(ws), _ptr(perm), _ptr(out),
             _stream())
    raster_perm.last_status = ws[:4].view(torch.int32)       # device status word: see check_raster_status
    return perm, out


def check_raster_status(status):
    """Raise when al3d_sp_raster_perm found its frame_rows_max promise broken (one small D2H: call it where the
    stream is synchronised anyway)."""
    v = int(status.item()) if status is not None else 0
    if v:
        raise lib.Al3dError("sparse encoder: frame_rows_max was promised (example['voxel_cap']) but "
                            + ("the voxel rows are not frame-sorted" if v & 1 else "a frame holds more than 65,535 voxels")
                            + "; pass frame_rows_max=0 for rows in any order")


def rows_gather_pad(rows, perm, channels_out, to_pair=False):
    """out[r] = rows[perm[r]] zero-padded to channels_out channels (f32 rows or pair rows)."""
    rows = _dev(rows, torch.float32, "rows")
    n, F = rows.shape
    out = torch.empty((n, channels_out), dtype=torch.float32, device=rows.device)
    lib.call("al3d_sp_rows_gather_pad_f32", _ptr(rows), _ptr(perm), n, F, channels_out, 1 if to_pair else 0, _ptr(out), _stream())
    return out


def tile_items(nbr, n_out, tmask):
    """Item list of a tiled 27-tap table: (first [ntiles+1] i32, items [9*ntiles+1, 4] i32)."""
    dev = nbr.device
    ntiles = (max(n_out, 1) + 31) // 32
    first = torch.empty((ntiles + 1,), dtype=torch.int32, device=dev)
    items = torch.empty((9 * ntiles + 1, 4), dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.load().al3d_sp_tile_items_workspace_bytes(n_out)), dtype=torch.uint8, device=dev)
    lib.call("al3d_sp_tile_items", _ptr(nbr), nbr.shape[1], 27, n_out, _ptr(tmask), _ptr(ws), _ptr(first), _ptr(items), _stream())
    return first, items


def sparse_glds(cin=None, cout=None):
    """True when the f16x3 sparse layer cin -> cout gets the LDS-DMA kernels' weight image (no arguments: any layer
    may): the per-tap gather kernel, or the range-gather kernel where sparse_rng says so."""
    if MATH != "f16x3" or SPCONV not in ("auto", "glds", "rng"):
        return False
    if cin is None or SPCONV == "glds":
        return True
    return (cin, cout) in GLDS_PAIRS or sparse_rng(cin, cout)


# ------------------------------------------------------------------ kernels
_DENSE_KIND = {torch.float32: "f32", torch.bfloat16: "bf16x6", torch.float16: "f16x3"}


def gap_fusable(w_packed):
    """The fused-GAP epilogue exists in the generic f16x3 kernels (plain f16 planes or LDS-DMA images: the deblock
    launches)."""
    if isinstance(w_packed, F16x3Packed):
        return w_packed.kind == "dma"
    return isinstance(w_packed, torch.Tensor) and w_packed.dtype == torch.float16


def conv2d_nhwc(x, w_packed, scale, shift, ksize, stride, pad, relu, out=None, coff=0, gap=None, io=0):
    """gap: optional [B, parts, ldc] f32 buffer (parts = al3d_gap_parts_count): the launch also writes its
    workgroups' channel sums there (f16x3 planes only, see gap_fusable)."""
    x = _dev(x, torch.float32, "x")
    # the weight format selects the arithmetic: bf16 [3,Cout,taps,Cin] (split_bf16x3),
    # f16 [2,Cout,taps,Cin] (split_f16x3, scale required) or plain f32 [Cout,taps,Cin]
    kind = _DENSE_KIND[w_packed.dtype]
    if kind == "f16x3" and scale is None:
        raise lib.Al3dError("conv2d_nhwc: f16x3 weights need the scale returned by split_f16x3")
    if isinstance(w_packed, F16x3Packed):             # fragment-ordered f16x3
        pk = w_packed
        B, H, W, Cin = x.shape
        if pk.cin != Cin or pk.taps != ksize * ksize or (pk.kind in ("frag3x3", "frag16", "wino") and (stride, pad) != (1, 1)):
            raise lib.Al3dError("conv2d_nhwc: fragment-ordered weights do not match this layer's geometry")
        OH = (H + 2 * pad - ksize) // stride + 1
        OW = (W + 2 * pad - ksize) // stride + 1
        if out is None:
            out = torch.empty((B, OH, OW, pk.cout), dtype=torch.float32, device=x.device)
        assert out.shape[:3] == (B, OH, OW) and out.is_contiguous()
        if pk.kind == "frag3x3":
            if io not in (0, IO_OUT_PAIR):
                raise lib.Al3dError("conv2d_nhwc: the streamed 3x3 kernel reads f32 pixels (it may write pair pixels)")
            if gap is not None:
                raise lib.Al3dError("conv2d_nhwc: this weight format has no fused GAP (see gap_fusable)")
            lib.call("al3d_conv3x3_nhwc_f16x3_frag_io", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift), _ptr(out),
                     B, H, W, Cin, pk.cout, out.shape[3], coff, 1 if relu else 0, io, _stream())
            return out
        if pk.kind == "wino":
            if io not in (0, IO_OUT_PAIR):
                raise lib.Al3dError("conv2d_nhwc: the Winograd 3x3 kernel reads f32 pixels (it may write pair pixels)")
            if gap is not None:
                raise lib.Al3dError("conv2d_nhwc: this weight format has no fused GAP (see gap_fusable)")
            lib.call("al3d_conv3x3_nhwc_f16x3_wino", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift), _ptr(out),
                     B, H, W, Cin, pk.cout, out.shape[3], coff, 1 if relu else 0, io, _stream())
            return out
        if pk.kind == "dma":
            lib.call("al3d_conv2d_nhwc_f16x3_dma", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift), _ptr(out),
                     B, H, W, Cin, pk.cout, ksize, stride, pad, out.shape[3], coff, 1 if relu else 0,
                     _ptr(gap), 0 if gap is None else gap.shape[1], io, _stream())
            return out
        if io:
            raise lib.Al3dError("conv2d_nhwc: pair pixels exist for the streamed 3x3 (output) and LDS-DMA kernels only")
        if gap is not None:
            raise lib.Al3dError("conv2d_nhwc: this weight format has no fused GAP (see gap_fusable)")
        if pk.kind == "frag16":
            lib.call("al3d_conv3x3_nhwc_f16x3_frag16", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift), _ptr(out),
                     B, H, W, Cin, pk.cout, out.shape[3], coff, 1 if relu else 0, _stream())
        else:
            lib.call("al3d_conv2d_nhwc_f16x3_bstream", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift),
                     _ptr(out), B, H, W, Cin, pk.cout, ksize, stride, pad, out.shape[3], coff,
                     1 if relu else 0, _stream())
        return out
    w_packed = _dev(w_packed, w_packed.dtype, "w")
    wshape = w_packed.shape[1:] if kind != "f32" else w_packed.shape
    B, H, W, Cin = x.shape
    Cout = wshape[0]
    assert wshape[1] == ksize * ksize and wshape[2] == Cin
    OH = (H + 2 * pad - ksize) // stride + 1
    OW = (W + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, OH, OW) and out.is_contiguous()
    if io:
        raise lib.Al3dError("conv2d_nhwc: pair pixels need fragment-ordered / LDS-DMA weights (pack_dense)")
    if gap is not None:
        if not gap_fusable(w_packed):
            raise lib.Al3dError("conv2d_nhwc: the fused GAP needs plain f16x3 planes")
        lib.call("al3d_conv2d_nhwc_f16x3_gap", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
                 B, H, W, Cin, Cout, ksize, stride, pad, out.shape[3], coff, 1 if relu else 0, _ptr(gap), gap.shape[1],
                 _stream())
        return out
    lib.call("al3d_conv2d_nhwc_" + kind, _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
             B, H, W, Cin, Cout, ksize, stride, pad, out.shape[3], coff, 1 if relu else 0, _stream())
    return out


def deconv2x2_nhwc(x, w_packed, scale, shift, relu, out=None, coff=0, gap=None, io=0):
    x = _dev(x, torch.float32, "x")
    B, H, W, Cin = x.shape
    kind = _DENSE_KIND[w_packed.dtype]
    if kind == "f16x3" and scale is None:
        raise lib.Al3dError("deconv2x2_nhwc: f16x3 weights need the scale returned by split_f16x3")
    if isinstance(w_packed, F16x3Packed):
        pk = w_packed
        if pk.kind not in ("bstream", "dma") or pk.taps != 4 or pk.cin != Cin:
            raise lib.Al3dError("deconv2x2_nhwc: fragment-ordered weights do not match this layer")
        if out is None:
            out = torch.empty((B, 2 * H, 2 * W, pk.cout), dtype=torch.float32, device=x.device)
        assert out.shape[:3] == (B, 2 * H, 2 * W) and out.is_contiguous()
        if pk.kind == "dma":
            lib.call("al3d_deconv2x2_nhwc_f16x3_dma", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift), _ptr(out),
                     B, H, W, Cin, pk.cout, out.shape[3], coff, 1 if relu else 0,
                     _ptr(gap), 0 if gap is None else gap.shape[1], io, _stream())
            return out
        if io:
            raise lib.Al3dError("deconv2x2_nhwc: pair pixels exist for the LDS-DMA kernel only")
        if gap is not None:
            raise lib.Al3dError("deconv2x2_nhwc: this weight format has no fused GAP (see gap_fusable)")
        lib.call("al3d_deconv2x2_nhwc_f16x3_bstream", _ptr(x), _ptr(pk.data), _ptr(scale), _ptr(shift),
                 _ptr(out), B, H, W, Cin, pk.cout, out.shape[3], coff, 1 if relu else 0, _stream())
        return out
    Cout = w_packed.shape[1] if kind != "f32" else w_packed.shape[0]
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, 2 * H, 2 * W) and out.is_contiguous()
    if gap is not None:
        if not gap_fusable(w_packed):
            raise lib.Al3dError("deconv2x2_nhwc: the fused GAP needs plain f16x3 planes")
        lib.call("al3d_deconv2x2_nhwc_f16x3_gap", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
                 B, H, W, Cin, Cout, out.shape[3], coff, 1 if relu else 0, _ptr(gap), gap.shape[1], _stream())
        return out
    lib.call("al3d_deconv2x2_nhwc_" + kind, _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
             B, H, W, Cin, Cout, out.shape[3], coff, 1 if relu else 0, _stream())
    return out


def gap_parts(OH, OW, deconv):
    return int(lib.load().al3d_gap_parts_count(int(OH), int(OW), 1 if deconv else 0))


def gap_reduce_parts(gap, count):
    """[B, parts, C] workgroup partial sums -> [B, C] means (sum in ascending part order / count)."""
    gap = _dev(gap, torch.float32, "gap")
    B, parts, C = gap.shape
    out = torch.empty((B, C), dtype=torch.float32, device=gap.device)
    lib.call("al3d_gap_reduce_parts_f32", _ptr(gap), B, parts, C, int(count), _ptr(out), _stream())
    return out


# AL3D_GAP=fused (default): the neck's deblock launches emit the embedding's partial sums; "kernel": the stand-alone
# two-stage GAP kernel re-reads the map (round-1 path; W-then-H summation order)
GAP = _os.environ.get("AL3D_GAP", "fused")


def gap_nhwc(x):
    x = _dev(x, torch.float32, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws = torch.empty((B, H, C), dtype=torch.float32, device=x.device)
    lib.call("al3d_gap_nhwc_f32", _ptr(x), B, H, W, C, _ptr(out), _ptr(ws), _stream())
    return out


# ------------------------------------------------------------------ voxelizer
class Voxelizer:
    """Batched device voxelizer + mean VFE (owns the persistent first-index grid).

    ``cfg`` keys follow the reference's ``voxel_generator`` dict
    (examples/active/cbgs_spatial_temporal.py:279-284): range, voxel_size,
    max_points_in_voxel, max_voxel_num.
    """

    def __init__(self, point_cloud_range, voxel_size, max_points_in_voxel, max_voxel_num,
                 max_batch=8, device="cuda"):
        import ctypes
        import numpy as np
        self.device = torch.device(device)
        rng = np.asarray(point_cloud_range, dtype=np.float32)
        vs = np.asarray(voxel_size, dtype=np.float32)
        # grid_size = round((max - min) / voxel_size) in float32 (voxel_generator.py:11-13)
        grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
        self.grid_size = grid                      # (x, y, z)
        self.range_min = (ctypes.c_float * 3)(*rng[:3].tolist())
        self.voxel_size = (ctypes.c_float * 3)(*vs.tolist())
        self.grid_c = (ctypes.c_int * 3)(*[int(g) for g in grid])
        self.max_points = int(max_points_in_voxel)
        self.max_voxels = int(max_voxel_num)
        self.max_batch = int(max_batch)
        nbytes = lib.load().al3d_voxelize_grid_bytes(self.max_batch, *[int(g) for g in grid])
        self.grid = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        lib.call("al3d_voxelize_grid_init", _ptr(self.grid), self.max_batch,
                 int(grid[0]), int(grid[1]), int(grid[2]), _stream())

    def __call__(self, points, point_offsets, want_voxels=False):
        """points [P,F] f32 (frames concatenated), point_offsets [B+1] i64 (device).
        Returns dict(feat [M,F], coords [M,4] i32 (b,z,y,x), num_points [M] i32,
        num_voxels [B] i32, voxels [M,max_points,F] | None)."""
        points = _dev(points, torch.float32, "points")
        point_offsets = _dev(point_offsets, torch.int64, "point_offsets")
        B = point_offsets.numel() - 1
        if B > self.max_batch:
            raise lib.Al3dError(f"batch {B} exceeds max_batch {self.max_batch}")
        npts, F = points.shape
        dev = points.device
        rows = B * self.max_voxels
        ws = torch.empty(lib.load().al3d_voxelize_workspace_bytes(npts, B, self.max_voxels),
                         dtype=torch.uint8, device=dev)
        feat = torch.empty((rows, F), dtype=torch.float32, device=dev)
        coords = torch.empty((rows, 4), dtype=torch.int32, device=dev)
        num_points = torch.empty((rows,), dtype=torch.int32, device=dev)
        voxels = torch.empty((rows, self.max_points, F), dtype=torch.float32, device=dev) \
            if want_voxels else None
        num_voxels = torch.empty((B,), dtype=torch.int32, device=dev)
        row_base = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        lib.call("al3d_voxelize_mean_f32", _ptr(points), _ptr(point_offsets), npts, B, F,
                 self.range_min, self.voxel_size, self.grid_c, self.max_points, self.max_voxels,
                 _ptr(self.grid), _ptr(ws), _ptr(feat), _ptr(coords), _ptr(num_points), _ptr(voxels),
                 _ptr(num_voxels), _ptr(row_base), _stream())
        m = int(row_base[-1].item())            # one small D2H per batch
        return dict(feat=feat[:m], coords=coords[:m], num_points=num_points[:m],
                    num_voxels=num_voxels, row_base=row_base, voxel_cap=self.max_voxels,
                    voxels=None if voxels is None else voxels[:m])


# ------------------------------------------------------------------ single sparse conv layer
MFMA_PAIRS = {(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)}


def sparse_conv_layer(feats, coords, batch, in_shape, weight, ksize, stride, pad, subm,
                      scale=None, shift=None, residual=None, relu=False, mfma=None, io=0):
    """One spconv layer on device (building block of the encoder, also used by the tests).
    feats [n,Cin] f32, coords [n,4] i32 (b,z,y,x), weight [kz,ky,kx,Cin,Cout].
    Returns (fout [n_out,Cout], coords_out [n_out,4], out_shape)."""
    import ctypes
    import numpy as np
    dev = feats.device
    st = _stream()
    k = [int(v) for v in ksize]
    n, cin = feats.shape
    cout = weight.shape[-1]
    K = k[0] * k[1] * k[2]
    w = weight.reshape(K, cin, cout).contiguous().float()
    D_, H_, W_ = [int(v) for v in in_shape]
    grid_in = torch.full((batch * D_ * H_ * W_,), -1, dtype=torch.int32, device=dev)
    lib.call("al3d_sp_scatter_index", _ptr(coords), n, batch, D_, H_, W_, _ptr(grid_in), 1, st)
    if mfma is None:
        mfma = (cin, cout) in MFMA_PAIRS and ({"bf16x6": "wave2", "f16x3": "glds_f16x3" if sparse_glds(cin, cout) else "wave2_f16x3"}
                                               .get(sparse_math(), True))
    tiled = mfma in ("glds_f16x3", "wave2_f16x3_tiles", "rng_f16x3", "r16_f16x3", "blk_f16x3")     # pitched table + per-tile tap masks
    tmask = None
    if subm:
        if tiled:
            pitch = lib.load().al3d_sp_table_pitch(n)
            nbr = torch.empty((K, pitch), dtype=torch.int32, device=dev)
            tmask = torch.empty((pitch // 32,), dtype=torch.int32, device=dev)
            lib.call("al3d_sp_subm_table_tiles", _ptr(coords), n, batch, D_, H_, W_, _ptr(grid_in), k[0], k[1], k[2],
                     _ptr(nbr), pitch, _ptr(tmask), st)
        else:
            nbr = torch.empty((max(n, 1), K), dtype=torch.int32, device=dev)
            lib.call("al3d_sp_subm_table", _ptr(coords), n, batch, D_, H_, W_, _ptr(grid_in), k[0], k[1], k[2],
                     _ptr(nbr), st)
        ocoords, n_out, oshape = coords, n, [D_, H_, W_]
    else:
        s3, p3 = [int(v) for v in stride], [int(v) for v in pad]
        oshape = [(in_shape[d] + 2 * p3[d] - (k[d] - 1) - 1) // s3[d] + 1 for d in range(3)]
        grid_out = torch.full((batch * oshape[0] * oshape[1] * oshape[2],), -1, dtype=torch.int32, device=dev)
        cap = min(n * K, grid_out.numel())
        ocoords = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        I3 = ctypes.c_int * 3
        lib.call("al3d_sp_down_claim", _ptr(coords), n, I3(*k), I3(*s3), I3(*p3), batch, *oshape,
                 _ptr(grid_out), _ptr(ocoords), _ptr(counter), cap, st)
        n_out = int(counter.item())
        ocoords = ocoords[:n_out].contiguous()
        if tiled:
            pitch = lib.load().al3d_sp_table_pitch(n_out)
            nbr = torch.empty((K, pitch), dtype=torch.int32, device=dev)
            tmask = torch.empty((pitch // 32,), dtype=torch.int32, device=dev)
            lib.call("al3d_sp_down_table_tiles", _ptr(ocoords), n_out, I3(*k), I3(*s3), I3(*p3), batch, D_, H_, W_,
                     _ptr(grid_in), _ptr(nbr), pitch, _ptr(tmask), st)
        else:
            nbr = torch.empty((max(n_out, 1), K), dtype=torch.int32, device=dev)
            lib.call("al3d_sp_down_table", _ptr(ocoords), n_out, I3(*k), I3(*s3), I3(*p3), batch, D_, H_, W_,
                     _ptr(grid_in), _ptr(nbr), st)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    if io and mfma not in ("glds_f16x3", "wave2_f16x3_tiles", "rng_f16x3", "r16_f16x3", "blk_f16x3"):
        raise lib.Al3dError("sparse_conv_layer: pair rows exist for the tiled f16x3 kernels only")
    if mfma == "glds_f16x3":
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        pk = pack_glds_f16x3(w3)
        lib.call("al3d_sp_conv_glds_f16x3_io", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), K, _ptr(pk.data), cin,
                 cout, _ptr(sc3), _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, io, st)
    elif mfma == "rng_f16x3":
        if not subm or K != 27:
            raise lib.Al3dError("sparse_conv_layer: the range-gather kernel serves 27-tap submanifold layers")
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        pk = pack_glds_f16x3(w3)
        trng = torch.empty((nbr.shape[1] // 32, 9, 2), dtype=torch.int32, device=dev)
        lib.call("al3d_sp_tile_ranges", _ptr(nbr), nbr.shape[1], K, n_out, _ptr(trng), st)
        lib.call("al3d_sp_conv_rng_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), _ptr(trng), K,
                 _ptr(pk.data), cin, cout, _ptr(sc3), _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, io, st)
    elif mfma == "blk_f16x3":
        if not subm or K != 27:
            raise lib.Al3dError("sparse_conv_layer: the block-staged kernel serves 27-tap submanifold layers")
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        pk = pack_glds_f16x3(w3)
        plan = block_plan(nbr, n_out, cin, cout)
        sparse_conv_layer.last_plan = plan
        lib.call("al3d_sp_conv_blk_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), _ptr(plan.hdr), _ptr(plan.rows),
                 _ptr(plan.loc), K, _ptr(pk.data), cin, cout, _ptr(sc3), _ptr(shift), _ptr(residual), 1 if relu else 0,
                 _ptr(out), n_out, io, st)
    elif mfma == "r16_f16x3":
        if K != 27 or cin != 16:
            raise lib.Al3dError("sparse_conv_layer: the item-stream kernel serves 27-tap layers with 16 input channels")
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        pk = pack_r16_f16x3(w3)
        first, items = tile_items(nbr, n_out, tmask)
        lib.call("al3d_sp_conv_r16_f16x3", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(items), _ptr(first), K, _ptr(pk.data),
                 cin, cout, _ptr(sc3), _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, io, R16_TPW, st)
    elif mfma == "wave2_f16x3_tiles":
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        lib.call("al3d_sp_conv_wave2_f16x3_tiles_io", _ptr(feats), _ptr(nbr), nbr.shape[1], _ptr(tmask), K, _ptr(w3), cin,
                 cout, _ptr(sc3), _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, io, st)
    elif mfma == "wave2_f16x3":
        w3, sc3 = split_f16x3(w.permute(2, 0, 1).contiguous(), scale)
        lib.call("al3d_sp_conv_wave2_f16x3", _ptr(feats), _ptr(nbr), K, _ptr(w3), cin, cout, _ptr(sc3),
                 _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, st)
    elif mfma in ("bf16x6", "wave", "wave2"):
        w6 = split_bf16x3(w.permute(2, 0, 1).contiguous())
        fn = {"bf16x6": "al3d_sp_conv_bf16x6", "wave": "al3d_sp_conv_wave_bf16x6",
              "wave2": "al3d_sp_conv_wave2_bf16x6"}[mfma]
        lib.call(fn, _ptr(feats), _ptr(nbr), K, _ptr(w6), cin, cout, _ptr(scale),
                 _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, st)
    elif mfma:
        w_ock = w.permute(2, 0, 1).contiguous()          # [Cout, K, Cin]
        lib.call("al3d_sp_conv_mfma_f32", _ptr(feats), _ptr(nbr), K, _ptr(w_ock), cin, cout, _ptr(scale),
                 _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, st)
    else:
        lib.call("al3d_sp_conv_f32", _ptr(feats), _ptr(nbr), K, _ptr(w), cin, cout, _ptr(scale),
                 _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(out), n_out, st)
    return out, ocoords, oshape


def box_decode(enc, anchors):
    enc = _dev(enc, torch.float32, "enc").reshape(-1, 10)
    anchors = _dev(anchors, torch.float32, "anchors").reshape(-1, 9)
    out = torch.empty((enc.shape[0], 9), dtype=torch.float32, device=enc.device)
    lib.call("al3d_box_decode_f32", _ptr(enc), _ptr(anchors), enc.shape[0], _ptr(out), _stream())
    return out
