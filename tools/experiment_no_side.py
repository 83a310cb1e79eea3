"""Dev experiment: how fast is the main stream when NOTHING runs beside it?  Convolves one prepared batch (voxels +
rulebook built once) 40 times on the main stream and compares with the pipelined sweep of 40 real batches: the
difference is what the side stream's voxelizer + rulebook cost the convolutions (contention), an upper bound for
what a cheaper index path could return."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import synthetic
from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
from al3d.models import build_detector
from al3d.sweep import sweep_embeddings
from al3d.utils import Config
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(root, "examples/active/cbgs_spatial_temporal_feature.py"))
model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
synthetic.seeded_init_(model, seed=0); model = model.to(dev).eval()
anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
B, NB = 64, 40
pool = PoolFrames.from_synthetic(B * NB, dev, num_base=16, seed=1000)
loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=B, device=dev)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sweep_embeddings(model, loader, dev, num_frames=len(pool))
    torch.cuda.synchronize(); t_sweep = time.perf_counter() - t0
print(f"pipelined sweep: {t_sweep / NB * 1e3:.2f} ms per batch  ({B * NB / t_sweep:.0f} frames/s)")
ex = next(iter(loader))
with torch.no_grad():
    book = model.prepare(ex)
    def once():
        bk = dict(book); bk["dense"] = torch.zeros_like(book["dense"])
        preds, middle = model(ex, return_loss=False, estimate=True, book=bk)
        return middle[-1].mean(-1).mean(-1)
    once(); torch.cuda.synchronize()
    for mode in ("with NMS side stream", ):
        t0 = time.perf_counter()
        for _ in range(NB): e = once()
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(f"main stream alone ({mode}): {t / NB * 1e3:.2f} ms per batch  ({B * NB / t:.0f} frames/s)")
    def once_no_nms():
        bk = dict(book); bk["dense"] = torch.zeros_like(book["dense"])
        x, middle = model.sparse_stage(ex, book=bk)
        x = model.neck(x)
        model.bbox_head(x)
        from al3d import detector_ops as DD
        return DD.gap_nhwc(x)
    once_no_nms(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(NB): e = once_no_nms()
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"main stream alone, no decode/NMS at all: {t / NB * 1e3:.2f} ms per batch  ({B * NB / t:.0f} frames/s)")
    # which part of the side work hurts?  Re-run the main-stream loop with ONE kind of side work per batch beside it
    side = torch.cuda.Stream(device=dev)
    frames = [pool.frames[i] for i in range(B)]
    off = torch.tensor([0] + list(torch.tensor([f.shape[0] for f in frames]).cumsum(0)), dtype=torch.int64, device=dev)
    pts = pool.flat[pool.offsets[0]:pool.offsets[B]]
    def side_vox():
        loader.voxelizer(pts, off)
    def side_book():
        model.prepare(ex)
    def side_tables_l0():
        bb = model.backbone
        from al3d import lib
        from al3d.selector_ops import _ptr, _stream
        c = ex["coordinates"]; n = c.shape[0]
        lv = bb._level([41, 1024, 1024], B, dev)
        lib.call("al3d_sp_scatter_index", _ptr(c), n, B, lv.D, lv.H, lv.W, _ptr(lv.grid), 1, _stream())
        nbr = torch.empty((27, n), dtype=torch.int32, device=dev)
        lib.call("al3d_sp_subm_table", _ptr(c), n, B, lv.D, lv.H, lv.W, _ptr(lv.grid), 3, 3, 3, _ptr(nbr), _stream())
        lib.call("al3d_sp_scatter_index", _ptr(c), n, B, lv.D, lv.H, lv.W, _ptr(lv.grid), 0, _stream())
    for name, fn in (("voxelizer", side_vox), ("rulebook (all levels)", side_book), ("level-0 grid + table only", side_tables_l0)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NB):
            with torch.cuda.stream(side):
                fn()
            e = once()
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        # the side work alone
        t1 = time.perf_counter()
        for _ in range(NB): fn()
        torch.cuda.synchronize(); ts = time.perf_counter() - t1
        print(f"main + side [{name}]: {t / NB * 1e3:.2f} ms per batch; that side work alone: {ts / NB * 1e3:.2f} ms")
