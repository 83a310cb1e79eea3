import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from al3d.utils import Config
from al3d.models import build_detector
from al3d import synthetic
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
from al3d.sweep import sweep_embeddings
dev = torch.device('cuda:0')
cfg = Config.fromfile('/root/repo/examples/active/cbgs_spatial_temporal_feature.py')
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0)
model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.time()
pool = PoolFrames.from_synthetic(nf, dev, num_base=4)
print('pool built', time.time() - t0, 's; pts/frame', pool.frames[0].shape)
loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)
with torch.no_grad():
    for it, ex in enumerate(loader):
        torch.cuda.synchronize(); t0 = time.time()
        preds, middle = model(ex, return_loss=False, estimate=True)
        emb = middle[-1].mean(dim=-1).mean(dim=-1)
        torch.cuda.synchronize()
        print(it, 'voxels', ex['coordinates'].shape[0], 'stage rows', [m.features.shape[0] for m in middle[:4]],
              'emb', tuple(emb.shape), float(emb.abs().mean()), 'dets', [p['box3d_lidar'].shape[0] for p in preds],
              f'{(time.time()-t0)*1e3:.1f} ms')
t0 = time.time()
E = sweep_embeddings(model, loader, dev, num_frames=nf)
torch.cuda.synchronize()
print('sweep', nf, 'frames', time.time() - t0, 's', E.shape, bool(torch.isfinite(E).all()))
