"""Dev tool: the sparse encoder layer by layer under the level-0 row orders (AL3D_L0 / R16_COUTS), on one real batch.

  python tools/bench_l0.py [batch] [iters]

Prints per layer the average time of its launch in each mode, the rulebook time, and checks that the dense output has the
same bits in all modes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import detector_ops as D, synthetic
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
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
enc = model.backbone
orig = B._SparseEncoderBase._conv
modes = [("off", "off", {16}, "f32"), ("raster16", "raster", {16}, "f32"), ("raster16+32", "raster", {16, 32}, "f32"),
         ("raster16p", "raster", {16}, "pair"), ("raster16+32p", "raster", {16, 32}, "pair")]
if os.environ.get("BENCH_L0_MODES"):
    modes = [m for m in modes if m[0] in os.environ["BENCH_L0_MODES"].split(",")]
results, ref = {}, None
for tag, l0, couts, rows in modes:
    D.L0, D.R16_COUTS, D.L0_ROWS = l0, couts, rows
    enc._packed_dev = None
    times, names = [], []
    def timed(m, feats, nbr, K, step, residual, out, n, st, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(m, feats, nbr, K, step, residual, out, n, st, **kw)
        e1.record()
        times.append((e0, e1))
        names.append(f"{feats.shape[-1]:3d}->{m.out_channels:3d} K={K:2d} n={n:8d} {'res' if residual is not None else '   '} {type(step['w']).__name__[:10]}")
    B._SparseEncoderBase._conv = staticmethod(timed)
    book_ms = []
    with torch.no_grad():
        for it in range(iters + 1):
            times.clear(); names.clear()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            book = enc.rulebook_for(ex["coordinates"], bs, ex["shape"][0], frame_rows_max=ex.get("voxel_cap", 0))
            b1.record()
            dense, _ = enc(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0], book=book)
            torch.cuda.synchronize()
            if it:
                book_ms.append(b0.elapsed_time(b1))
                acc = [a + e0.elapsed_time(e1) for a, (e0, e1) in zip(acc, times)]
            else:
                acc = [0.0] * len(times)
    results[tag] = ([a / iters * 1e3 for a in acc], list(names), sum(book_ms) / len(book_ms))
    if ref is None:
        ref = dense.clone()
    else:
        print(tag, "dense output bit-identical to", modes[0][0], ":", bool(torch.equal(ref.view(torch.int32), dense.view(torch.int32))),
              " max |diff| / max |ref| =", float((dense - ref).abs().max() / ref.abs().max()))
B._SparseEncoderBase._conv = staticmethod(orig)
tags = [m[0] for m in modes]
print("layer".ljust(44), *[t.rjust(14) for t in tags])
for i, nm in enumerate(results[tags[0]][1]):
    print(nm.ljust(44), *[f"{results[t][0][i]:14.1f}" for t in tags])
print("sum us per batch".ljust(44), *[f"{sum(results[t][0]):14.1f}" for t in tags])
print("rulebook ms (host-timed, incl. syncs)".ljust(44), *[f"{results[t][2]:14.3f}" for t in tags])
