"""Dev probe: how sparse are the rulebooks really (valid pairs vs non-empty 32-row sub-tiles)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
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
bs = 4
pool = PoolFrames.from_synthetic(bs, dev, num_base=4)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
with torch.no_grad():
    x, middle = model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
levels = [("L0", ex["coordinates"], [41, 1024, 1024])] + [(f"L{i+1}", m.indices, m.spatial_shape) for i, m in enumerate(middle[:3])]
for name, coords, shape in levels:
    n = coords.shape[0]
    D, H, W = shape
    grid = torch.full((bs * D * H * W,), -1, dtype=torch.int32, device=dev)
    def table(c):
        g = grid.clone()
        lib.call("al3d_sp_scatter_index", _ptr(c), n, bs, D, H, W, _ptr(g), 1, _stream())
        nbr = torch.empty((27, n), dtype=torch.int32, device=dev)
        lib.call("al3d_sp_subm_table", _ptr(c), n, bs, D, H, W, _ptr(g), 3, 3, 3, _ptr(nbr), _stream())
        return nbr
    def stats(nbr, tag):
        valid = (nbr >= 0)
        pad = (-n) % 32
        v = torch.nn.functional.pad(valid, (0, pad)).view(27, -1, 32).any(-1)
        v16 = torch.nn.functional.pad(valid, (0, (-n) % 16)).view(27, -1, 16).any(-1)
        v128 = torch.nn.functional.pad(valid, (0, (-n) % 128)).view(27, -1, 128).any(-1)
        print(f"{name} {tag:10s} n={n:7d} valid pairs {valid.float().mean():.3f}  non-empty 16-row x tap {v16.float().mean():.3f}  32-row x tap {v.float().mean():.3f}  128-row x tap {v128.float().mean():.3f}")
    c = coords.contiguous()
    stats(table(c), "as-is")
    key = ((c[:, 0].long() * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]
    stats(table(c[torch.argsort(key)].contiguous()), "raster")
    # z-innermost order: (b, y, x, z)
    key2 = ((c[:, 0].long() * H + c[:, 2]) * W + c[:, 3]) * D + c[:, 1]
    stats(table(c[torch.argsort(key2)].contiguous()), "yxz")
    # 2-D blocked (8x8 in y,x) then z
    key3 = (((c[:, 0].long() * (H // 8 + 1) + c[:, 2] // 8) * (W // 8 + 1) + c[:, 3] // 8) * D + c[:, 1]) * 64 + (c[:, 2] % 8) * 8 + c[:, 3] % 8
    stats(table(c[torch.argsort(key3)].contiguous()), "blk8x8,z")
    # orderings by the row's own tap mask (the executed (32-row, tap) units only depend on WHICH rows share a tile, so
    # permuting the table's columns is enough to evaluate them): full 27-bit mask sort (ideal grouping), the six face
    # neighbours as a 6-bit bucket key (a counting sort), popcount buckets
    nbr = table(c)
    valid = (nbr >= 0)
    w = (1 << torch.arange(27, device=dev, dtype=torch.int64)).view(27, 1)
    mask = (valid.long() * w).sum(0)
    face = [4, 10, 12, 14, 16, 22]                                     # (kz,ky,kx) = centre +- one axis, k = (kz*3+ky)*3+kx
    fkey = sum(((mask >> k) & 1) << i for i, k in enumerate(face))
    pop = valid.sum(0)
    for tag, key in (("mask-sort", mask), ("face6 bkt", fkey), ("popcount", pop), ("pop,mask", pop * (1 << 27) + mask)):
        perm = torch.argsort(key, stable=True)
        stats(nbr[:, perm], tag)

    # the same mask sort inside WINDOWS of consecutive raster rows (keeps a tile's rows spatial neighbours): executed fraction by window size
    nbr_r = table(c[torch.argsort(((c[:, 0].long() * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3])].contiguous())
    valid_r = (nbr_r >= 0)
    mask_r = (valid_r.long() * w).sum(0)
    for win in (64, 128, 256, 512, 1024, 4096):
        idx = torch.arange(n, device=dev)
        key = (idx // win) * (1 << 28) + mask_r
        stats(nbr_r[:, torch.argsort(key, stable=True)], f"mask/w{win}")
        # cheaper key: popcount of the mask's three kz groups + ky groups (what makes whole (kz,ky) groups live)
