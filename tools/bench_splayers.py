"""Dev tool: A/B the sparse-conv kernels layer by layer on the real rulebooks of one batch.

  python tools/bench_splayers.py [batch] [fnA,fnB,...]

Runs the sparse encoder once with the layer calls recorded (AL3D_MATH selects the weight format:
bf16x6 entries need AL3D_MATH=bf16x6), then replays every matrix-core layer with
each of the named C-ABI entry points (same arguments), checks the outputs agree bit for bit with
the first one, and prints the time per layer and the total."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import detector_ops as D
D.SPCONV = "wave2" if D.MATH == "f16x3" else D.SPCONV      # record plain f16 planes; the glds image is packed below
from al3d import lib, synthetic
from al3d.selector_ops import _ptr, _stream
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
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fns = (sys.argv[2] if len(sys.argv) > 2 else "al3d_sp_conv_wave2_f16x3").split(",")
pool = PoolFrames.from_synthetic(bs, dev, num_base=8)
ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=bs, device=dev)))
calls = []
orig = B._SparseEncoderBase._conv
def rec(m, feats, nbr, K, step, residual, out, n, st, tmask=None, trng=None, io=0):
    calls.append((m, feats, nbr, K, step, residual, n, io))
    return orig(m, feats, nbr, K, step, residual, out, n, st, tmask=tmask, trng=trng, io=io)
B._SparseEncoderBase._conv = staticmethod(rec)
with torch.no_grad():
    model.backbone(ex["voxel_features"], ex["coordinates"], bs, ex["shape"][0])
torch.cuda.synchronize()
tot = {f: 0.0 for f in fns}
tiled_cache = {}
for (m, feats, nbr, K, step, residual, n, io) in calls:
    if step["w"].dtype not in (torch.bfloat16, torch.float16):
        continue
    ci, co = m.in_channels, m.out_channels
    valid = float((nbr[:, :n] >= 0).float().mean()) if nbr.dim() == 2 else -1
    line = f"{ci:3d}->{co:3d} K={K:2d} n={n:7d} valid={valid:.2f} "
    ref = None
    ci = feats.shape[-1]                                    # 16 for the zero-padded first layer
    plain = nbr if nbr.shape[1] == n else nbr[:, :n].contiguous()      # the encoder records pitched (tiled) tables
    tiled = None
    if True:                                                # the tiled form of this table: pitch + tile masks (+ ranges)
        key = nbr.data_ptr()
        if key not in tiled_cache:
            pitch = lib.load().al3d_sp_table_pitch(n)
            tn = torch.full((K, pitch), -1, dtype=torch.int32, device=dev)
            tn[:, :n] = nbr[:, :n]
            tm = ((tn.view(K, pitch // 32, 32) >= 0).any(-1).to(torch.int64) << torch.arange(K, device=dev)[:, None]).sum(0).to(torch.int32)
            tr = torch.zeros((pitch // 32, 9, 2), dtype=torch.int32, device=dev)
            if K == 27:
                lib.call("al3d_sp_tile_ranges", _ptr(tn), pitch, K, n, _ptr(tr), _stream())
            tiled_cache[key] = (tn, tm.contiguous(), pitch, tr)
        tiled = tiled_cache[key]
    for f in fns:
        out = torch.empty((n, co), device=dev)
        w = D.pack_glds_f16x3(step["w"]).data if ("glds" in f or "rng" in f) else step["w"]
        def call():
            if "rng" in f:
                if not m.subm:
                    raise lib.Al3dError("submanifold only")
                lib.call(f, _ptr(feats), _ptr(tiled[0]), tiled[2], _ptr(tiled[1]), _ptr(tiled[3]), K, _ptr(w), ci, co,
                         _ptr(step["scale"]), _ptr(step["shift"]), None if residual is None else _ptr(residual), 1, _ptr(out),
                         n, io, _stream())
                return
            if f.endswith("_io"):                           # glds / wave2-on-tiles with the recorded row formats
                lib.call(f, _ptr(feats), _ptr(tiled[0]), tiled[2], _ptr(tiled[1]), K, _ptr(w), ci, co, _ptr(step["scale"]),
                         _ptr(step["shift"]), None if residual is None else _ptr(residual), 1, _ptr(out), n, io, _stream())
                return
            if "glds" in f:
                lib.call(f, _ptr(feats), _ptr(tiled[0]), tiled[2], _ptr(tiled[1]), K, _ptr(w), ci, co, _ptr(step["scale"]),
                         _ptr(step["shift"]), None if residual is None else _ptr(residual), 1, _ptr(out), n, _stream())
                return
            lib.call(f, _ptr(feats), _ptr(plain), K, _ptr(w), ci, co, _ptr(step["scale"]), _ptr(step["shift"]),
                     None if residual is None else _ptr(residual), 1, _ptr(out), n, _stream())
        try:
            call()
        except lib.Al3dError as e:
            line += f"| {f[13:]:>12s}: n/a "
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(ref, out))                  # the kernel structures share one arithmetic: same bits
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        tot[f] += us
        line += f"| {f[13:]:>12s}: {us:7.1f} us {2.0*n*K*ci*co/us/1e6:6.1f} TF {'=' if same else 'DIFF'} "
    print(line)
print("total us per batch:", {f[13:]: round(v, 1) for f, v in tot.items()}, " per frame ms:", {f[13:]: round(v / bs / 1e3, 4) for f, v in tot.items()})
