"""Dev tool: time the rulebook kernels (sp_subm_table per level) on one real batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import lib, synthetic
from al3d.selector_ops import _ptr, _stream
from al3d.utils import Config
from al3d.models import build_detector
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device('cuda:0')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, 'examples/active/cbgs_spatial_temporal_feature.py'))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0); model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
with torch.no_grad():
    x, middle = model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
levels = [("L0", ex["coordinates"], [41, 1024, 1024])] + [(f"L{i+1}", m.indices, m.spatial_shape) for i, m in enumerate(middle[:3])]
tot = 0.0
for name, coords, shape in levels:
    c = coords.contiguous()
    n = c.shape[0]
    D, H, W = shape
    grid = torch.full((bs * D * H * W,), -1, dtype=torch.int32, device=dev)
    lib.call("al3d_sp_scatter_index", _ptr(c), n, bs, D, H, W, _ptr(grid), 1, _stream())
    nbr = torch.empty((27, n), dtype=torch.int32, device=dev)
    def call():
        lib.call("al3d_sp_subm_table", _ptr(c), n, bs, D, H, W, _ptr(grid), 3, 3, 3, _ptr(nbr), _stream())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tot += us
    print(f"{name} n={n:8d} subm_table {us:8.1f} us  checksum {int(nbr.sum(dtype=torch.int64))}")
print(f"total {tot:.1f} us per batch of {bs}")
