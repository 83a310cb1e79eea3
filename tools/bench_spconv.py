"""Micro-benchmark of the sparse conv kernels on synthetic rulebooks (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import lib
from al3d.selector_ops import _ptr, _stream
dev = "cuda:0"
def run(cin, cout, n, K, mode, fill):
    g = torch.Generator(device="cpu").manual_seed(0)
    fin = torch.randn(n, cin, device=dev)
    w = torch.randn(cout, K, cin, device=dev) * 0.05
    if mode == "random":
        nbr = torch.randint(0, n, (K, n), generator=g, dtype=torch.int32)
    else:   # coherent: neighbour of row r at tap k is row r + k - K/2
        nbr = (torch.arange(n, dtype=torch.int32)[None, :] + (torch.arange(K, dtype=torch.int32)[:, None] - K // 2)).clamp(0, n - 1)
    if fill < 1.0:
        drop = torch.rand(K, n, generator=g) > fill
        nbr = torch.where(drop, torch.full_like(nbr, -1), nbr)
    nbr = nbr.contiguous().to(dev)
    out = torch.empty(n, cout, device=dev)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    def call():
        lib.call("al3d_sp_conv_mfma_f32", _ptr(fin), _ptr(nbr), K, _ptr(w), cin, cout, _ptr(sc), _ptr(sh), None, 1, _ptr(out), n, _stream())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * n * K * cin * cout
    print(f"{cin:3d}->{cout:3d} n={n:6d} K={K:2d} {mode:8s} fill={fill:.2f}: {us:8.1f} us  {fl/us/1e6:7.1f} TF(dense-eq)")
for (ci, co, n) in [(64, 64, 62000), (128, 128, 58000), (32, 32, 170000), (16, 16, 480000)]:
    for mode in ("coherent", "random"):
        for fill in (1.0, 0.3):
            run(ci, co, n, 27, mode, fill)
    run(ci, co, n, 1, "coherent", 1.0)
