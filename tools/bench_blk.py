"""Dev tool (GPU box): per-layer times of the sparse encoder, serial (no side stream), with the block-staged kernel
off / on, plus the plan's statistics.   python tools/bench_blk.py [batch] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import detector_ops as D
from al3d import synthetic
from al3d.utils import Config
from al3d.models import build_detector
from al3d.datasets import generate_task_anchors, PoolFrames, DeviceSweepLoader
dev = torch.device('cuda:0')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, 'examples/active/cbgs_spatial_temporal_feature.py'))
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["off", "on"]
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
saved = set(D.BLK_PAIRS) or set(D.BLK_BUILT)        # "on" = every built pair unless AL3D_BLK_PAIRS names some
res = {}
for mode in modes:
    D.BLK_PAIRS = set() if mode == "off" else saved
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0); model = model.to(dev).eval()
    bb = model.backbone
    ev = []
    conv = bb._conv
    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(*a, **k); e1.record()
        ev.append((a[0].in_channels, a[0].out_channels, a[3], e0, e1))
    bb._conv = timed
    with torch.no_grad():
        book = bb.rulebook_for(ex["coordinates"], bs, ex["shape"][0])
        for b in book["steps"]:
            p = b.get("plan")
            if p is not None and not getattr(p, "_seen", False):
                p._seen = True
                h = p.hdr.cpu().numpy()
                print(f"[{mode}] plan R={p.R} cap={p.cap}: chunks {h.shape[0]}, per-tap {int(h[:,1].sum())}, staged rows mean {h[h[:,1]==0,0].mean():.1f} "
                      f"p99 {sorted(h[:,0])[int(0.99*len(h))]} max {h[:,0].max()}  amp {h[:,0].sum()/max(1,b['n']):.2f}")
        for r in range(reps + 1):
            if r == 1: ev.clear()
            bk = dict(book); bk["dense"] = torch.zeros_like(book["dense"])
            out, _ = bb(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0], book=bk)
    torch.cuda.synchronize()
    nl = len(ev) // reps
    us = [0.0] * nl
    for i, (ci, co, K, e0, e1) in enumerate(ev):
        us[i % nl] += e0.elapsed_time(e1) * 1e3 / reps
    res[mode] = (us, [(e[0], e[1], e[2]) for e in ev[:nl]], out)
for i in range(len(res[modes[0]][0])):
    ci, co, K = res[modes[0]][1][i]
    print(f"{ci:3d}->{co:3d} K={K:2d} " + "  ".join(f"{m}: {res[m][0][i]:8.1f} us" for m in modes))
print("total ms:", {m: round(sum(res[m][0]) / 1e3, 3) for m in modes})
if len(modes) == 2:
    print("same bits:", bool(torch.equal(res[modes[0]][2], res[modes[1]][2])))
