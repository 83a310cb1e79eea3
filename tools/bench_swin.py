"""Dev tool: Swin-T (BASELINE configs[4] shape: 6 cameras of 256 x 704 per sample) on the token kernels -- ms per sample.

  python tools/bench_swin.py [samples=2] [reps=5]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from al3d.models.swin import SwinTransformer
from al3d.synthetic import seed_modules_

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
swin = seed_modules_(SwinTransformer(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4,
                                     qkv_bias=True, patch_norm=True, out_indices=[1, 2, 3]), 23).to(dev)
img = torch.randn(S * 6, 256, 704, 3, device=dev)
with torch.no_grad():
    for _ in range(2):
        swin(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        swin(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
gflop = 194.0 * S
print(f"swin-t {S} samples ({S * 6} images): {dt * 1e3:.2f} ms per forward = {dt * 1e3 / S:.2f} ms per sample, "
      f"{gflop / dt / 1e3:.1f} TFLOP/s algorithmic")
