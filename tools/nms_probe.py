"""Dev tool: time MultiGroupHead.predict (score pre-pass + decode/NMS kernel) on one batch of 32."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from al3d import synthetic
from al3d.utils import Config
from al3d.models import build_detector
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(ROOT, 'examples/active/cbgs_spatial_temporal_feature.py'))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0); model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
pool = PoolFrames.from_synthetic(32, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=32, device=dev)))
with torch.no_grad():
    x, middle = model.sparse_stage(ex)
    x = model.neck(x)
    preds = model.bbox_head(x)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = model.bbox_head.predict(ex, preds, model.test_cfg)
        _ = out[0]["scores"]
        torch.cuda.synchronize(); print("predict ms", round((time.time() - t0) * 1e3, 2))
