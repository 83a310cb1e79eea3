"""Dev probe (GPU box): executed vs valid work of every sparse layer of the headline encoder: valid = (row, tap) pairs with a
neighbour / (rows x taps); live = (32-row tile, tap) pairs the kernels execute / (tiles x taps), from the tiled rulebook's masks."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from al3d import synthetic
from al3d.utils import Config
from al3d.models import build_detector
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, "examples/active/cbgs_spatial_temporal_feature.py"))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0)
model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pool = PoolFrames.from_synthetic(bs, dev, num_base=16)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
bb = model.backbone
with torch.no_grad():
    book = bb.build_rulebook(ex["coordinates"], bs, np.array(ex["shape"][0][::-1]) + [1, 0, 0])
steps = [s for s in book["steps"] if isinstance(s, dict) and "nbr" in s]
plan = [s for s in bb._plan if s["kind"] != "stage_end"]
assert len(steps) == len(plan), (len(steps), len(plan))
for st, pl in zip(steps, plan):
    m = pl["mod"]
    nbr, n, K, tmask = st["nbr"], st["n"], st["K"], st["tmask"]
    valid = float((nbr[:, :n] >= 0).float().mean())
    if tmask is None:
        print(f"{pl['kind']:5s} {m.in_channels:3d}->{m.out_channels:3d} K={K} n={n} valid {valid:.3f} (untiled)")
        continue
    nt = (n + 31) // 32
    tm = tmask[:nt].cpu().numpy().astype(np.uint32)
    live = sum(int(((tm >> k) & 1).sum()) for k in range(K)) / (nt * K)
    # 16-row granularity from the table itself
    v = (nbr[:, :n] >= 0)
    pad = (-n) % 16
    v16 = torch.nn.functional.pad(v, (0, pad)).view(K, -1, 16).any(-1).float().mean().item()
    print(f"{pl['kind']:5s} {m.in_channels:3d}->{m.out_channels:3d} K={K:2d} n={n:8d} valid {valid:.3f} live32 {live:.3f} live16 {v16:.3f} "
          f"executed/valid {live / valid:.2f}")
