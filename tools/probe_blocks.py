"""Dev probe (GPU box): 3-D blocked staging statistics of the sparse encoder's levels on one real batch.
For block shapes (BZ, BY, BX): rows a block owns, active cells in its halo box (= rows a workgroup would stage),
amplification = staged / owned, 32-row tile fill."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from al3d import synthetic
from al3d.utils import Config
from al3d.models import build_detector
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device('cuda:0')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, 'examples/active/cbgs_spatial_temporal_feature.py'))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0); model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
bb = model.backbone
with torch.no_grad():
    book = bb.rulebook_for(ex["coordinates"], bs, ex["shape"][0])
levels = [(ex["coordinates"].int(), [41, 1024, 1024])]
for step, b in zip(bb._plan, book["steps"]):
    if step["kind"] == "stage_end":
        levels.append((b["coords"], b["shape"]))
# levels[0] = input level (level 0); stage_end k holds the coords AFTER stage k's strided conv = level k+1
shapes = [(1, 1, 32), (1, 2, 16), (2, 2, 8), (2, 4, 8), (2, 4, 16), (4, 4, 8), (4, 4, 16), (2, 8, 8), (4, 8, 8), (2, 8, 16), (4, 8, 16), (8, 8, 8)]
for li, (coords, shape) in enumerate(levels[:4]):
    Dz, H, W = [int(s) for s in shape]
    n = coords.shape[0]
    occ = torch.zeros((bs, 1, Dz, H, W), dtype=torch.float32, device=dev)
    c = coords.long()
    occ[c[:, 0], 0, c[:, 1], c[:, 2], c[:, 3]] = 1.0
    print(f"level {li}: shape {shape} rows {n} ({n / bs:.0f} per frame)")
    for (bz, by, bx) in shapes:
        pz, py, px = (-Dz) % bz, (-H) % by, (-W) % bx
        o = F.pad(occ, (0, px, 0, py, 0, pz))
        own = F.avg_pool3d(o, (bz, by, bx), (bz, by, bx)) * (bz * by * bx)
        o2 = F.pad(o, (1, 1, 1, 1, 1, 1))
        staged = F.avg_pool3d(o2, (bz + 2, by + 2, bx + 2), (bz, by, bx)) * ((bz + 2) * (by + 2) * (bx + 2))
        own, staged = own.flatten().round(), staged.flatten().round()
        m = own > 0
        own, staged = own[m], staged[m]
        tiles = torch.ceil(own / 32)
        q = torch.quantile(own, torch.tensor([0.5, 0.9, 0.99], device=dev))
        qs = torch.quantile(staged, torch.tensor([0.5, 0.9, 0.99], device=dev))
        print(f"  block {bz}x{by}x{bx:2d} ({bz*by*bx:4d} cells): blocks {int(m.sum()):7d}  own mean {own.mean():6.1f} p50 {q[0]:4.0f} p90 {q[1]:4.0f} p99 {q[2]:4.0f} max {int(own.max()):4d} | "
              f"staged mean {staged.mean():6.1f} p50 {qs[0]:4.0f} p90 {qs[1]:4.0f} p99 {qs[2]:4.0f} max {int(staged.max()):4d} | amp {staged.sum() / own.sum():.2f}  tile fill {own.sum() / (32 * tiles.sum()):.3f}  fill(cells) {own.sum()/(m.sum()*bz*by*bx):.3f}")
    del occ
