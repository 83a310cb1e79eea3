"""GPU suite: rows of a sparse level re-numbered by their tap masks inside windows of raster rows (al3d_sp_mask_window_sort).
Integer work is exact against numpy; the encoder on re-numbered levels equals the raster encoder bit for bit."""
import numpy as np
import pytest
import torch

from test_detector_oracle import random_sparse

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _masks(coords, batch, shape):
    """27-bit submanifold neighbour mask per row, numpy."""
    D, H, W = shape
    occ = np.zeros((batch, D + 2, H + 2, W + 2), dtype=bool)
    occ[coords[:, 0], coords[:, 1] + 1, coords[:, 2] + 1, coords[:, 3] + 1] = True
    m = np.zeros(coords.shape[0], dtype=np.int64)
    for k in range(27):
        kz, ky, kx = k // 9, (k // 3) % 3, k % 3
        m |= occ[coords[:, 0], coords[:, 1] + kz, coords[:, 2] + ky, coords[:, 3] + kx].astype(np.int64) << k
    return m


@pytest.mark.parametrize("window", [1024, 4096, 8192, 16384])
@pytest.mark.parametrize("shape,batch,n", [([9, 40, 37], 3, 6001), ([5, 21, 19], 1, 40), ([11, 64, 48], 2, 3000),
                                           ([21, 100, 90], 2, 40000)])
def test_mask_window_sort_is_a_stable_sort_by_mask_inside_windows(shape, batch, n, window):
    from al3d import lib
    from al3d.selector_ops import _ptr, _stream
    rng = np.random.default_rng(n + window)
    _, coords = random_sparse(rng, batch, shape, n, 4)
    D_, H_, W_ = shape
    key = ((coords[:, 0].astype(np.int64) * D_ + coords[:, 1]) * H_ + coords[:, 2]) * W_ + coords[:, 3]
    coords = coords[np.argsort(key, kind="stable")]                       # raster order, as al3d_sp_down_sites leaves a level
    grid = torch.full((batch * D_ * H_ * W_,), -1, dtype=torch.int32, device=DEV)
    c = _t(coords)
    lib.call("al3d_sp_scatter_index", _ptr(c), n, batch, D_, H_, W_, _ptr(grid), 1, _stream())
    out = torch.full((n, 4), -9, dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.load().al3d_sp_mask_window_sort_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    lib.call("al3d_sp_mask_window_sort", _ptr(c), n, batch, D_, H_, W_, _ptr(grid), window, _ptr(out), _ptr(ws), _stream())
    got = out.cpu().numpy()
    masks = _masks(coords, batch, shape)
    want = np.empty_like(coords)
    for b0 in range(0, n, window):
        sl = slice(b0, min(b0 + window, n))
        want[sl] = coords[sl][np.argsort(masks[sl], kind="stable")]
    assert np.array_equal(got, want)
    g = grid.cpu().numpy().reshape(batch, D_, H_, W_)
    assert np.array_equal(g[got[:, 0], got[:, 1], got[:, 2], got[:, 3]], np.arange(n))
    assert (g >= 0).sum() == n


def test_mask_window_sort_rejects_bad_arguments():
    from al3d import lib
    from al3d.selector_ops import _ptr, _stream
    c = torch.zeros((4, 4), dtype=torch.int32, device=DEV)
    g = torch.full((18,), -1, dtype=torch.int32, device=DEV)
    ws = torch.empty(256, dtype=torch.uint8, device=DEV)
    with pytest.raises(lib.Al3dError, match="window"):
        lib.call("al3d_sp_mask_window_sort", _ptr(c), 4, 1, 2, 3, 3, _ptr(g), 100, _ptr(torch.empty_like(c)), _ptr(ws), _stream())
    with pytest.raises(lib.Al3dError, match="in-place"):
        lib.call("al3d_sp_mask_window_sort", _ptr(c), 4, 1, 2, 3, 3, _ptr(g), 1024, _ptr(c), _ptr(ws), _stream())


@pytest.mark.parametrize("window", [1024, 16384])
@pytest.mark.parametrize("widths", [{128}, {32, 64, 128}])
def test_encoder_on_mask_sorted_levels_equals_the_raster_encoder(widths, window):
    """The whole encoder with levels re-numbered by tap mask == the encoder on raster levels, bit for bit through the dense BEV
    map: every kernel sums a row's taps in tap order, whatever the row's number is.  (Width 32 re-numbers level 1, whose layers
    then leave the range-gather kernel's fast path: still the same bits.)"""
    from al3d import detector_ops as D, synthetic
    from al3d.models.backbones import FPNSpMiddleResNetFHD
    rng = np.random.default_rng(5)
    batch, n = 2, 9000
    feats, coords = random_sparse(rng, batch, [40, 96, 88], n, 5)
    outs = []
    saved = (set(D.MASK_SORT), D.MASK_SORT_WINDOW)
    try:
        for ms in (set(), widths):
            D.MASK_SORT, D.MASK_SORT_WINDOW = ms, window
            m = FPNSpMiddleResNetFHD(num_input_features=5)
            synthetic.seeded_init_(m, seed=0)
            m = m.to(DEV).eval()
            with torch.no_grad():
                dense, middle = m(_t(feats), _t(coords), batch, [88, 96, 40])
            outs.append(dense.cpu().numpy())
    finally:
        D.MASK_SORT, D.MASK_SORT_WINDOW = saved
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    assert np.array_equal(outs[0].view(np.int32), outs[1].view(np.int32))


def test_full_size_detector_is_bit_identical_with_and_without_the_mask_sort():
    """BASELINE configs[1] size: three synthetic 10-sweep frames through the shipped detector (voxelizer, encoder, neck, head)
    with levels 2-3 grouped by tap mask and in raster order: the dense BEV map, the embedding and the head's raw output are
    the same bits -- every kernel sums a row's taps in tap order, whatever the row's number."""
    import os
    from al3d import detector_ops as D, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.utils import Config
    from test_detector_oracle import G
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active", "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_synthetic(3, DEV, num_base=3)
    outs = []
    saved = set(D.MASK_SORT)
    try:
        for ms in (set(), {64, 128}):
            D.MASK_SORT = ms
            ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=3, device=DEV)))
            with torch.no_grad():
                bev, _ = model.backbone(ex["voxel_features"], ex["coordinates"], 3, ex["shape"][0])
                dets, middle = model(ex, return_loss=False, estimate=True)
            emb = middle[-1].mean(-1).mean(-1) if hasattr(middle[-1], "mean") else None
            outs.append((bev.cpu().numpy(), None if emb is None else emb.cpu().numpy(),
                         [d["box3d_lidar"].cpu().numpy() for d in dets], [d["scores"].cpu().numpy() for d in dets]))
    finally:
        D.MASK_SORT = saved
    a, b = outs
    assert np.abs(a[0]).max() > 0
    assert np.array_equal(a[0].view(np.int32), b[0].view(np.int32))
    if a[1] is not None:
        assert np.array_equal(a[1].view(np.int32), b[1].view(np.int32))
    for x, y in zip(a[2] + a[3], b[2] + b[3]):
        assert np.array_equal(x.view(np.int32), y.view(np.int32))
