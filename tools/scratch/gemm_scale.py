import os, sys
sys.path.insert(0, "/root/repo")
import torch
from al3d import token_ops as T
dev = "cuda:0"
shapes = []
for mult in (4, 16, 64):
    shapes += [("s2 qkv x%d" % mult, 24696 * mult // 4 * 1, 384, 1152), ("s2 fc1 x%d" % mult, 16896 * mult // 4, 384, 1536), ("s2 fc2 x%d" % mult, 16896 * mult // 4, 1536, 384)]
shapes += [("big K", 65536, 4096, 1024), ("big all", 131072, 1024, 1024), ("N128", 262144, 1024, 128)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    pk = T.PackedLinear(w, torch.zeros(N, device=dev))
    out = torch.empty(M, N, device=dev)
    for _ in range(2):
        T.linear(a, pk, a_pair=True, out=out, out_pair=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        T.linear(a, pk, a_pair=True, out=out, out_pair=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    print(f"{name:12s} M={M:7d} K={K:4d} N={N:4d}: {us:8.1f} us  {2.0*M*K*N/us/1e6:6.1f} TFLOP/s", flush=True)
