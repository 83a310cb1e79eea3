"""Dev tool: the assembled BEVFusion camera+lidar model (BASELINE configs[4] shapes: 6 cameras 256 x 704, 0.075 m voxels,
1440 x 1440 x 41 grid) on synthetic inputs with seeded weights -- per-stage time and frames/s of the embedding sweep.

  python tools/bench_bevfusion_camera_lidar.py [batch=2] [reps=3] [head=0]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from al3d import synthetic
from al3d.datasets import DeviceSweepLoader, PoolFrames
from al3d.models import build_detector
from al3d.models.bevfusion_model import BEVFusionCameraLidar, transfusion_head_for
from al3d.utils import Config
from al3d.synthetic import camera_setup as _camera_setup, seed_modules_ as _seed_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with_head = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "examples", "active", "bevfusion_lidar_spatial_temporal_feature.py"))
lidar = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(lidar, seed=0)
model = BEVFusionCameraLidar(lidar, head=transfusion_head_for() if with_head else None)
for i, m in enumerate((model.camera_backbone, model.camera_neck, model.vtransform, model.fuser) + ((model.head,) if with_head else ())):
    _seed_(m, 30 + i)
model = model.to(dev).eval()
pool = PoolFrames.from_synthetic(B, dev, num_base=min(B, 4), seed=1)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, None, B, device=dev)))
K, cam2lidar, lidar2image, img_aug, lidar_aug, _ = _camera_setup(B, 6, 9, (256, 704))
img = torch.randn(B, 6, 256, 704, 3, device=dev)
points = [pool.frames[i] for i in range(B)]
args = (ex, img, points, lidar2image.to(dev), K.to(dev), cam2lidar.to(dev), img_aug.to(dev), lidar_aug.to(dev))
with torch.no_grad():
    emb, dec, preds = model(*args, timed=True)                         # warm-up (packs weights)
    assert emb.shape == (B, 512) and bool(torch.isfinite(emb).all()), emb.shape
    emb, dec, preds = model(*args, timed=True)
    for k, v in model.stage_ms.items():
        print(f"{k:32s} {v / B:8.2f} ms per sample")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print(f"batch {B}: {dt * 1e3:.1f} ms per batch = {B / dt:.1f} frames/s (camera+lidar embedding{' + head' if with_head else ''}); "
      f"voxels/frame {int(ex['num_voxels'][0])}, decoder map {tuple(dec.shape)}, boxes {None if preds is None else len(preds[0]['scores'])}")
