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
import json
# algorithmic HBM bytes per launch of the kernels that serve ONE shape here (tools/rocpd_summary.py hbm reads this line):
# the fused halves read the residual stream once and write it once
t0_, t1_ = S * 6 * 64 * 176, S * 6 * 32 * 88
print(json.dumps({"algorithmic_mb": {"tok_attn_block_f16x3_kernel<96,": t0_ * 96 * 8 / 1e6,
                                     "tok_attn_block_f16x3_kernel<192,": t1_ * 192 * 8 / 1e6,
                                     "tok_mlp_f16x3_kernel<96,": t0_ * 96 * 8 / 1e6,
                                     "tok_mlp16_f16x3_kernel<192>": t1_ * 192 * 8 / 1e6}}))
print(f"swin-t {S} samples ({S * 6} images): {dt * 1e3:.2f} ms per forward = {dt * 1e3 / S:.2f} ms per sample, "
      f"{gflop / dt / 1e3:.1f} TFLOP/s algorithmic")
