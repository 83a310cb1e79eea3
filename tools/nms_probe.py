"""Dev tool: time MultiGroupHead.predict (score pre-pass + decode/NMS kernel) on one batch (argv[1] frames, default 32)."""
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
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pool = PoolFrames.from_synthetic(NB, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=NB, device=dev)))
with torch.no_grad():
    x, middle = model.sparse_stage(ex)
    x = model.neck(x)
    preds = model.bbox_head(x)
    out = model.bbox_head.predict(ex, preds, model.test_cfg)        # creates the head's side stream
    side = model.bbox_head._side
    torch.cuda.synchronize()
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        out = model.bbox_head.predict(ex, preds, model.test_cfg)
        e1.record(side)
        torch.cuda.synchronize()
        print("score pre-pass + decode/NMS kernels: %.3f ms (batch %d)" % (e0.elapsed_time(e1), NB))
