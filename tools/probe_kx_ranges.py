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
def rec(m, feats, nbr, K, step, residual, out, n, st, **kw):
    calls.append((m, nbr, K, n))
    return orig(m, feats, nbr, K, step, residual, out, n, st, **kw)
B._SparseEncoderBase._conv = staticmethod(rec)
with torch.no_grad():
    model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
seen = set()
# variant: 32-row tiles, groups of the three kx taps of one (kz, ky): contiguous range [lo, hi] of their valid neighbours
for (m, nbr, K, n) in calls:
    key = (m.in_channels, m.out_channels, K, n, m.subm)
    if key in seen or K != 27 or not m.subm:
        continue
    seen.add(key)
    t = nbr[:, :n] if nbr.dim() == 2 else nbr.view(K, -1)[:, :n]
    R = 32
    nt = (n + R - 1) // R
    tt = torch.nn.functional.pad(t, (0, nt * R - n), value=-1).view(9, 3, nt, R)
    valid = tt >= 0
    lo = torch.where(valid, tt, torch.full_like(tt, 2 ** 30)).amin(dim=(1, 3))          # [9, nt]
    hi = torch.where(valid, tt, torch.full_like(tt, -1)).amax(dim=(1, 3))
    live = hi >= 0
    L = torch.clamp(hi - lo + 1, min=0)[live].float()
    refs = valid.sum(dim=(1, 3))[live].float()
    taps_live = valid.any(dim=3).sum(dim=1)[live].float()                                # live taps per live group
    q = torch.quantile(L, torch.tensor([0.5, 0.9, 0.99], device=dev))
    print(f"{m.in_channels:3d}->{m.out_channels:3d} n={n:7d}: live groups {live.float().mean():.3f}  L median {q[0]:.0f} p90 {q[1]:.0f} "
          f"p99 {q[2]:.0f} mean {L.mean():.1f}  <=32 {(L <= 32).float().mean():.3f} <=48 {(L <= 48).float().mean():.3f} "
          f"<=64 {(L <= 64).float().mean():.3f}  refs/group {refs.mean():.1f}  live taps/group {taps_live.mean():.2f}  "
          f"rows now (32 x live taps) {32 * taps_live.mean():.0f}")
