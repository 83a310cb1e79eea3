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


# ------------------------------------------------------------------ kernels
def conv2d_nhwc(x, w_packed, scale, shift, ksize, stride, pad, relu, out=None, coff=0):
    x = _dev(x, torch.float32, "x")
    w_packed = _dev(w_packed, torch.float32, "w")
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    assert w_packed.shape[1] == ksize * ksize and w_packed.shape[2] == Cin
    OH = (H + 2 * pad - ksize) // stride + 1
    OW = (W + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, OH, OW) and out.is_contiguous()
    lib.call("al3d_conv2d_nhwc_f32", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
             B, H, W, Cin, Cout, ksize, stride, pad, out.shape[3], coff, 1 if relu else 0, _stream())
    return out


def deconv2x2_nhwc(x, w_packed, scale, shift, relu, out=None, coff=0):
    x = _dev(x, torch.float32, "x")
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    assert out.shape[:3] == (B, 2 * H, 2 * W) and out.is_contiguous()
    lib.call("al3d_deconv2x2_nhwc_f32", _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(out),
             B, H, W, Cin, Cout, out.shape[3], coff, 1 if relu else 0, _stream())
    return out


def gap_nhwc(x):
    x = _dev(x, torch.float32, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    lib.call("al3d_gap_nhwc_f32", _ptr(x), B, H, W, C, _ptr(out), _stream())
    return out
