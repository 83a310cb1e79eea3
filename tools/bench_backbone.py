"""Dev tool: run only voxelizer + sparse encoder on one batch repeatedly (for rocprof)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d.utils import Config
from al3d.models import build_detector
from al3d import synthetic
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device('cuda:0')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, 'examples/active/cbgs_spatial_temporal_feature.py'))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0)
model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)
ex = next(iter(loader))
with torch.no_grad():
    for it in range(6):
        x, middle = model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
    torch.cuda.synchronize()
print("rows", [m.features.shape[0] for m in middle])
