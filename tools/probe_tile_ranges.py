"""Dev probe: for the SubM layers of one real batch, how many distinct input rows does a tile of R
consecutive output rows reference if, per kz-group of 9 taps, the contiguous index range
[min valid nbr, max valid nbr] is staged (rows are in raster order from level 1 on, so a tap's
neighbour index is monotone in the output row)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import synthetic
from al3d.utils import Config
from al3d.models import build_detector
from al3d.models import backbones as B
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
calls = []
orig = B._SparseEncoderBase._conv
def rec(m, feats, nbr, K, step, residual, out, n, st):
    calls.append((m, nbr, K, n))
    return orig(m, feats, nbr, K, step, residual, out, n, st)
B._SparseEncoderBase._conv = staticmethod(rec)
with torch.no_grad():
    model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
seen = set()
for (m, nbr, K, n) in calls:
    key = (m.in_channels, m.out_channels, K, n, m.subm)
    if key in seen or K != 27 or not m.subm:
        continue
    seen.add(key)
    t = nbr[:, :n] if nbr.dim() == 2 else nbr.view(K, -1)[:, :n]
    mono = 0
    for k in (0, 13, 26):
        v = t[k][t[k] >= 0]
        mono += int((v[1:] < v[:-1]).sum())
    for R in (64, 128, 256):
        nt = (n + R - 1) // R
        pad = nt * R - n
        tt = torch.nn.functional.pad(t, (0, pad), value=-1).view(27, nt, R)
        tot = torch.zeros(nt, dtype=torch.int64, device=dev)
        for g in range(3):
            grp = tt[9 * g:9 * g + 9]                                  # [9, nt, R]
            valid = grp >= 0
            lo = torch.where(valid, grp, torch.full_like(grp, 2 ** 30)).amin(dim=(0, 2))
            hi = torch.where(valid, grp, torch.full_like(grp, -1)).amax(dim=(0, 2))
            tot += torch.clamp(hi - lo + 1, min=0) * (hi >= 0)
        refs = (tt >= 0).sum(dim=(0, 2)).float()
        q = torch.quantile(tot.float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
        print(f"{m.in_channels:3d}->{m.out_channels:3d} n={n:7d} R={R:3d}: staged rows/tile median {q[0]:.0f} p90 {q[1]:.0f} "
              f"p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {int(tot.max())}  refs/tile {refs.mean():.0f}  "
              f"frac tiles <= 1.5R+64: {(tot <= 1.5 * R + 64).float().mean():.3f}  <= 3R+48: {(tot <= 3 * R + 48).float().mean():.3f}  non-monotone steps {mono}")
