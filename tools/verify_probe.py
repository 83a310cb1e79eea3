# exercise the sampled verification branch on one GPU with a 6400-frame metadata pool and synthetic feats
import sys, os, random
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from al3d import synthetic, selector_ops as ops
infos, logs = synthetic.make_pool(160, seed=0)
n = len(infos)
dev = torch.device('cuda:0')
feats = torch.from_numpy(synthetic.make_embeddings(n, seed=2, scale=0.01)).to(dev)
cfgm, run_id, n_boxes = synthetic.pool_arrays(infos)
xy = np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfgm])
S = ops.spatial_map(torch.from_numpy(xy).to(dev), 8)
F = ops.l1_distance(feats, 2)
D = ops.combine_maps(n, spatial=S, temporal_id=torch.from_numpy(run_id).to(dev), feat=F)
random.seed(3407); first = random.choice(range(n))
rc, picks = ops.greedy_kcenter(D, [], first, torch.from_numpy(n_boxes * 0.04).to(dev), 0.12, 0.0, 600.0)
import time; t0 = time.time()
print('verify (sampled branch):', bench.verify_selection(infos, feats, picks), f'{time.time()-t0:.1f}s', len(picks))
