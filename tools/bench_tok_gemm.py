"""Dev tool: the token GEMM on the Swin-T layer shapes (4 samples = 24 images of 256 x 704).
  python tools/bench_tok_gemm.py     (the ablation numbers in DESIGN.md 5.3 came from dev builds of csrc/tokens.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import token_ops as T
dev = "cuda:0"
shapes = [("s0 qkv", 305760, 96, 288, False), ("s0 proj", 305760, 96, 96, False), ("s0 fc1", 270336, 96, 384, True),
          ("s0 fc2", 270336, 384, 96, False), ("s1 qkv", 76440, 192, 576, False), ("s1 fc1", 67584, 192, 768, True),
          ("s1 fc2", 67584, 768, 192, False), ("s2 qkv", 24696, 384, 1152, False), ("s2 fc2", 16896, 1536, 384, False),
          ("s3 qkv", 9408, 768, 2304, False), ("s3 fc2", 4224, 3072, 768, False)]
for name, M, K, N, pair_out in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    pk = T.PackedLinear(w, torch.zeros(N, device=dev))
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        T.linear(a, pk, a_pair=True, out=out, out_pair=pair_out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        T.linear(a, pk, a_pair=True, out=out, out_pair=pair_out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:8s} M={M:6d} K={K:4d} N={N:4d}: {us:7.1f} us  {2.0*M*K*N/us/1e6:6.1f} TFLOP/s  rows {(M*K+M*N)*4/us/1e3:6.0f} GB/s")
