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
