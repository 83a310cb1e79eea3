"""Dev probe: error of the f16x3 conv (fraction of sum|a*b|) against fp64 over activation magnitudes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from al3d import detector_ops as D

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
B, Cin, H, W, Cout, k = 1, 128, 32, 32, 128, 3
base = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g))
w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
wp = D.pack_conv_weight(w).to(DEV)
w3, sc3 = D.split_f16x3(wp)
w6 = D.split_bf16x3(wp)
for mag in (1e3, 1.0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    x = base * mag
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    ref = F.conv2d(x.double(), w.double(), padding=1).permute(0, 2, 3, 1)
    sc = F.conv2d(x.abs().double(), w.abs().double(), padding=1).permute(0, 2, 3, 1)
    e = {}
    e["f32"] = ((D.conv2d_nhwc(xn, wp, None, None, k, 1, 1, False).cpu().double() - ref).abs() / sc).max().item()
    e["bf16x6"] = ((D.conv2d_nhwc(xn, w6, None, None, k, 1, 1, False).cpu().double() - ref).abs() / sc).max().item()
    e["f16x3"] = ((D.conv2d_nhwc(xn, w3, sc3, None, k, 1, 1, False).cpu().double() - ref).abs() / sc).max().item()
    print(f"mag {mag:8.0e}: " + "  ".join(f"{n} {v:.3e}" for n, v in e.items()))
