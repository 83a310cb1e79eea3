"""Dev probe: selector-stage timings and APSP correctness at multi-GPU pool sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from al3d import synthetic, selector_ops as ops
import oracle
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
infos, logs = synthetic.make_pool(scenes, seed=0)
cfgm, run_id, n_boxes = synthetic.pool_arrays(infos)
xy = np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfgm])
n = len(infos)
feats = torch.from_numpy(synthetic.make_embeddings(n, seed=1, scale=0.01)).to(dev)
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize()
    print(f"{name:12s} {1e3*(time.time()-t0):9.2f} ms"); return r
xyd = torch.from_numpy(xy).to(dev)
for rep in range(2):
    d, i = timed("knn", lambda: ops.knn_2d(xyd, 9))
    S = timed("apsp", lambda: ops.apsp_knn(d, i))
    F = timed("l1", lambda: ops.l1_distance(feats, 2))
    D = timed("combine", lambda: ops.combine_maps(n, spatial=S, temporal_id=torch.from_numpy(run_id).to(dev), feat=F))
    box = torch.from_numpy(n_boxes * 0.04).to(dev)
    rc, picks = timed("greedy", lambda: ops.greedy_kcenter(D, [], 232, box, 0.12, 0.0, 600.0))
    print("picks", len(picks), rc)
# correctness of a few APSP rows against the oracle (works for any n)
kd, ki = oracle.knn(xy, 9)
indptr, indices, w = oracle.knn_csr(kd, ki)
rows = [0, 1, n // 2, n - 1]
Sc = S.cpu().numpy()
ok = True
for r in rows:
    ref = oracle.apsp(indptr, indices, w, r, r + 1)[0]
    ok &= np.array_equal(ref.view(np.int64), Sc[r].view(np.int64))
print("n", n, "apsp rows bit-exact vs oracle:", ok, "finite frac", float(np.isfinite(Sc[0]).mean()))
