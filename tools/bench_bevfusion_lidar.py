"""Dev tool: embedding-sweep throughput of the BEVFusion lidar branch config (BASELINE configs[3]:
0.075 m voxels, 1440 x 1440 x 41 grid, 160k-voxel cap, no detection head) on one GPU.

  python tools/bench_bevfusion_lidar.py [frames=320] [batch=16]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from al3d import sweep as S, synthetic
from al3d.datasets import DeviceSweepLoader, PoolFrames
from al3d.models import build_detector
from al3d.utils import Config

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0)
model = model.to(dev).eval()
pool = PoolFrames.from_synthetic(n, dev, num_base=16, seed=1)
loader = DeviceSweepLoader(pool, cfg.voxel_generator, None, batch, device=dev)
ex = next(iter(loader))
print(f"voxels per frame: {[int(v) for v in ex['num_voxels'][:4]]} ...  grid {list(ex['shape'][0])}")
del ex
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = S.sweep_embeddings(model, loader, dev, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sweep {it}: {n} frames in {dt:.3f} s = {n / dt:.1f} frames/s  (finite: {bool(torch.isfinite(e).all())}, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)")
