"""GPU suite: the level-0 path of the sparse encoder (csrc/spconv_l0.hip) -- raster renumbering of the voxelizer's rows,
item lists, and the item-stream kernel for 16 input channels.  Integer work (permutation, coordinates, items) is
checked exactly against numpy; the convolution is bit-identical to the register-gather kernel (same MFMA sequence per
output row), which tests/test_detector_gpu.py checks against the oracle (spconv itself absent: parity unpinned)."""
import numpy as np
import pytest
import torch

from test_detector_oracle import random_sparse

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _raster_key(coords, shape):
    c = coords.astype(np.int64)
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


@pytest.mark.parametrize("case", ["sparse", "long_lines", "wide", "single", "empty", "one_line_full"])
def test_raster_perm_equals_a_stable_sort(case):
    """al3d_sp_raster_perm: perm / coords in raster order == numpy's argsort of the (b, z, y, x) key (cells are unique, so
    the order is total).  Cases: scattered voxels; lines with hundreds of members (the bit-mask rank's multi-pass loop);
    W = 1440 (more than 32 mask words, BEVFusion's grid); one voxel; none; one completely filled line."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(len(case))
    batch, shape = 3, [5, 40, 300]
    if case == "sparse":
        cells = rng.choice(batch * shape[0] * shape[1] * shape[2], size=7001, replace=False)
    elif case == "long_lines":
        shape = [3, 6, 1024]
        cells = rng.choice(batch * shape[0] * shape[1] * shape[2], size=30011, replace=False)
    elif case == "wide":
        batch, shape = 2, [2, 9, 1440]
        cells = rng.choice(batch * shape[0] * shape[1] * shape[2], size=20001, replace=False)
    elif case == "single":
        cells = np.array([12345])
    elif case == "empty":
        cells = np.zeros((0,), dtype=np.int64)
    else:
        batch, shape = 1, [2, 3, 2048]
        cells = (1 * 3 + 2) * 2048 + rng.permutation(2048)
    x = cells % shape[2]
    y = (cells // shape[2]) % shape[1]
    z = (cells // (shape[2] * shape[1])) % shape[0]
    b = cells // (shape[2] * shape[1] * shape[0])
    coords = np.stack([b, z, y, x], 1).astype(np.int32)
    coords = coords[rng.permutation(len(coords))]
    perm, cr = D.raster_perm(_t(coords).reshape(-1, 4), batch, shape)
    torch.cuda.synchronize()
    want = np.argsort(_raster_key(coords, shape), kind="stable")
    assert np.array_equal(perm.cpu().numpy(), want.astype(np.int32))
    assert np.array_equal(cr.cpu().numpy(), coords[want])


@pytest.mark.parametrize("case", ["frames", "one_frame_full_lines", "empty_frames", "cap_hit"])
def test_raster_perm_frame_sorted_fast_path(case):
    """Frame-sorted rows with a promised per-frame cap (the voxelizer's output): one workgroup per frame sorts in LDS
    (16-bit packed line counters, LDS scan, bit-mask rank over chunks of 512 lines).  Same contract: == the stable sort.
    Cases: several frames of different sizes; lines with hundreds of members crossing chunk borders; frames without a row
    in between; a frame at the 65,535-row limit."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(len(case) + 7)
    shape = [41, 96, 1024]
    if case == "frames":
        counts = [5001, 1, 12000, 333, 0, 7000]
    elif case == "one_frame_full_lines":
        shape = [3, 700, 1024]                                          # 2,100 lines: several chunks, dense lines
        counts = [60000]
    elif case == "empty_frames":
        counts = [0, 0, 4000, 0, 2500, 0]
    else:
        counts = [65535, 100]
    rows = []
    for b, c in enumerate(counts):
        cells = rng.choice(shape[0] * shape[1] * shape[2], size=c, replace=False)
        if case == "one_frame_full_lines":                                  # most rows in a few hundred completely filled lines
            cells = np.concatenate([np.arange(58 * 1024) + 1024 * 600, rng.choice(1024 * 600, size=c - 58 * 1024, replace=False)])
            cells = rng.permutation(cells)
        x, y, z = cells % shape[2], (cells // shape[2]) % shape[1], cells // (shape[2] * shape[1])
        rows.append(np.stack([np.full_like(x, b), z, y, x], 1))
    coords = np.concatenate(rows).astype(np.int32)
    batch = len(counts)
    perm, cr = D.raster_perm(_t(coords), batch, shape, frame_rows_max=max(counts))
    torch.cuda.synchronize()
    want = np.argsort(_raster_key(coords, shape), kind="stable")
    assert np.array_equal(perm.cpu().numpy(), want.astype(np.int32))
    assert np.array_equal(cr.cpu().numpy(), coords[want])
    assert int(D.raster_perm.last_status.item()) == 0                       # the promise was kept
    slow_perm, slow_cr = D.raster_perm(_t(coords), batch, shape)            # the general path gives the same answer
    assert torch.equal(slow_perm, perm) and torch.equal(slow_cr, cr)


@pytest.mark.parametrize("case", ["unsorted", "too_many"])
def test_raster_perm_checks_its_promise(case):
    """frame_rows_max is a promise the device CHECKS.  Rows that are not frame-sorted, or a frame with more
    than 65,535 rows: the status word is set, perm is still a permutation of all rows and coords_raster the matching
    coordinates (nothing uninitialised reaches a gather), and the encoder's check raises."""
    from al3d import detector_ops as D, lib
    rng = np.random.default_rng(11)
    shape = [41, 96, 1024]
    counts = [3000, 2000, 1000] if case == "unsorted" else [65536 + 500, 100]
    rows = []
    for b, c in enumerate(counts):
        cells = rng.choice(shape[0] * shape[1] * shape[2], size=c, replace=False)
        x, y, z = cells % shape[2], (cells // shape[2]) % shape[1], cells // (shape[2] * shape[1])
        rows.append(np.stack([np.full_like(x, b), z, y, x], 1))
    coords = np.concatenate(rows).astype(np.int32)
    if case == "unsorted":
        coords = coords[rng.permutation(len(coords))]
    perm, cr = D.raster_perm(_t(coords), len(counts), shape, frame_rows_max=60000)
    status = D.raster_perm.last_status
    torch.cuda.synchronize()
    assert int(status.item()) == (1 if case == "unsorted" else 2)
    p = perm.cpu().numpy()
    assert np.array_equal(np.sort(p), np.arange(len(coords)))
    assert np.array_equal(cr.cpu().numpy(), coords[p])
    with pytest.raises(lib.Al3dError):
        D.check_raster_status(status)


def test_rows_gather_pad():
    """out[r] = rows[perm[r]] zero-padded (5 -> 16 channels), as f32 rows and as pair rows (== rows_convert of the f32 form);
    perm = None is the identity."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(3)
    rows = rng.normal(size=(1003, 5)).astype(np.float32)
    rows[::5] *= 1e-4
    perm = rng.permutation(1003).astype(np.int32)
    got = D.rows_gather_pad(_t(rows), _t(perm), 16)
    want = np.zeros((1003, 16), np.float32)
    want[:, :5] = rows[perm]
    assert np.array_equal(got.cpu().numpy(), want)
    ident = D.rows_gather_pad(_t(rows), None, 16)
    assert np.array_equal(ident.cpu().numpy()[:, :5], rows) and not ident.cpu().numpy()[:, 5:].any()
    pair = D.rows_gather_pad(_t(rows), _t(perm), 16, to_pair=True)
    assert torch.equal(pair.view(torch.int32), D.rows_convert(got, True).view(torch.int32))
    wide = D.rows_gather_pad(_t(want), _t(perm), 16)                      # no padding: a pure gather
    assert np.array_equal(wide.cpu().numpy(), want[perm])


def _tiled_table(coords, batch, shape, strided):
    """(nbr [27, pitch], tmask, n_out, out coords) of a tiled 27-tap table through the C ABI."""
    import ctypes
    from al3d import lib
    from al3d.selector_ops import _ptr, _stream
    c = _t(coords)
    n = c.shape[0]
    D_, H_, W_ = shape
    grid = torch.full((batch * D_ * H_ * W_,), -1, dtype=torch.int32, device=DEV)
    lib.call("al3d_sp_scatter_index", _ptr(c), n, batch, D_, H_, W_, _ptr(grid), 1, _stream())
    if not strided:
        pitch = lib.load().al3d_sp_table_pitch(n)
        nbr = torch.empty((27, pitch), dtype=torch.int32, device=DEV)
        tmask = torch.empty((pitch // 32,), dtype=torch.int32, device=DEV)
        lib.call("al3d_sp_subm_table_tiles", _ptr(c), n, batch, D_, H_, W_, _ptr(grid), 3, 3, 3, _ptr(nbr), pitch, _ptr(tmask),
                 _stream())
        return nbr, tmask, n, c
    I3 = ctypes.c_int * 3
    oshape = [(shape[d] + 2 - 2 - 1) // 2 + 1 for d in range(3)]
    ogrid = torch.full((batch * oshape[0] * oshape[1] * oshape[2],), -1, dtype=torch.int32, device=DEV)
    cap = min(n * 27, ogrid.numel())
    oc = torch.empty((cap, 4), dtype=torch.int32, device=DEV)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.load().al3d_sp_down_sites_workspace_bytes(batch, *oshape), dtype=torch.uint8, device=DEV)
    lib.call("al3d_sp_down_sites", _ptr(c), n, I3(3, 3, 3), I3(2, 2, 2), I3(1, 1, 1), batch, *oshape, _ptr(ogrid), _ptr(oc),
             _ptr(counter), cap, _ptr(ws), _stream())
    n_out = int(counter.item())
    oc = oc[:n_out].contiguous()
    pitch = lib.load().al3d_sp_table_pitch(n_out)
    nbr = torch.empty((27, pitch), dtype=torch.int32, device=DEV)
    tmask = torch.empty((pitch // 32,), dtype=torch.int32, device=DEV)
    lib.call("al3d_sp_down_table_tiles", _ptr(oc), n_out, I3(3, 3, 3), I3(2, 2, 2), I3(1, 1, 1), batch, D_, H_, W_, _ptr(grid),
             _ptr(nbr), pitch, _ptr(tmask), _stream())
    return nbr, tmask, n_out, oc


@pytest.mark.parametrize("strided", [False, True])
@pytest.mark.parametrize("n", [17, 4001])
def test_tile_items_equal_the_table(strided, n):
    """al3d_sp_tile_items: per tile the live (kz, ky) groups in ascending order with the exact (min, max - min + 1) of the
    valid entries of their three kx columns, first / last flags, the tile's tap mask, the prefix table and the dummy."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(n + strided)
    batch, shape = 2, [7, 30, 41]
    _, coords = random_sparse(rng, batch, shape, n, 1)
    coords = coords[np.argsort(_raster_key(coords, shape), kind="stable")]
    nbr, tmask, n_out, _ = _tiled_table(coords, batch, shape, strided)
    first, items = D.tile_items(nbr, n_out, tmask)
    torch.cuda.synchronize()
    nb, tm, first, items = nbr.cpu().numpy(), tmask.cpu().numpy(), first.cpu().numpy(), items.cpu().numpy()
    ntiles = (n_out + 31) // 32
    k = 0
    for t in range(ntiles):
        assert first[t] == k
        live = [g for g in range(9) if (int(tm[t]) >> (3 * g)) & 7]
        assert live, "every tile has a live group (the centre tap / at least one input)"
        for i, g in enumerate(live):
            v = nb[3 * g:3 * g + 3, 32 * t:32 * t + 32]
            v = v[v >= 0]
            lo, meta, mask, tile = items[k]
            assert lo == v.min() and (meta & 0xffff) == v.max() - v.min() + 1
            assert (meta >> 16) & 15 == g and tile == t and mask == tm[t]
            assert bool(meta & (1 << 20)) == (i == 0) and bool(meta & (1 << 21)) == (i == len(live) - 1)
            k += 1
    assert first[ntiles] == k and tuple(items[k]) == (0, 0, 0, ntiles)


@pytest.mark.parametrize("cout", [16, 32])
@pytest.mark.parametrize("geom", ["subm", "down"])
@pytest.mark.parametrize("order", ["raster", "random", "tiny"])
def test_item_stream_kernel_is_bit_identical(cout, geom, order):
    """sp_conv_r16_kernel performs sp_conv_wave2's MFMA sequence per output row (taps ascending; xl' wd, xh wl, xh wh into
    one accumulator): same bits.  Raster order: staged ranges; random order: every range is long, the per-row fallback;
    ragged last tile; residual + ReLU (16 -> 16); strided table; inputs smaller than a tile; several tiles-per-wave
    settings (items streamed across tile boundaries)."""
    from al3d import detector_ops as D
    if geom == "down" and order == "tiny":
        pytest.skip("covered by the submanifold tiny case")
    rng = np.random.default_rng(cout + len(order) + len(geom))
    if order == "tiny":
        shape, batch, n = [3, 5, 4], 1, 17
    else:
        shape, batch, n = [9, 40, 37], 3, 6001
    feats, coords = random_sparse(rng, batch, shape, n, 16)
    if order != "random":
        perm = np.argsort(_raster_key(coords, shape), kind="stable")
        feats, coords = feats[perm], coords[perm]
    feats[::7] *= 1e-3
    feats[5::11] *= 300.0
    subm = geom == "subm"
    k, s, p = ((3, 3, 3), (1, 1, 1), (0, 0, 0)) if subm else ((3, 3, 3), (2, 2, 2), (1, 1, 1))
    w = (rng.normal(size=(3, 3, 3, 16, cout)) / np.sqrt(16 * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32) if (subm and cout == 16) else None

    def run(mode):
        got, gco, osh = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), k, s, p, subm, scale=_t(scale),
                                            shift=_t(shift), residual=None if res is None else _t(res), relu=True, mfma=mode)
        key = _raster_key(gco.cpu().numpy(), osh)                        # strided sites are claimed with atomics: align by cell
        return got.cpu().numpy()[np.argsort(key, kind="stable")]
    ref = run("wave2_f16x3_tiles")
    assert ref.shape[0] > 0 and np.isfinite(ref).all()
    saved = D.R16_TPW
    try:
        for tpw in (0, 1, 3, 64):
            D.R16_TPW = tpw
            got = run("r16_f16x3")
            assert np.array_equal(ref.view(np.int32), got.view(np.int32)), tpw
    finally:
        D.R16_TPW = saved
    if res is not None:                                                      # the form without a residual
        res = None
        assert np.array_equal(run("wave2_f16x3_tiles").view(np.int32), run("r16_f16x3").view(np.int32))


@pytest.mark.parametrize("cout", [16, 32])
def test_item_stream_kernel_pair_rows(cout):
    """Row formats (csrc/sp_rows.h) through the item-stream kernel: pair rows in, out and as the residual give the bits the
    register-gather kernel gives with the same flags."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(cout)
    shape, batch, n = [9, 40, 37], 3, 5003
    feats, coords = random_sparse(rng, batch, shape, n, 16)
    perm = np.argsort(_raster_key(coords, shape), kind="stable")
    feats, coords = feats[perm], coords[perm]
    feats[::7] *= 1e-3
    feats[5::11] *= 300.0
    w = (rng.normal(size=(3, 3, 3, 16, cout)) / np.sqrt(16 * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32) if cout == 16 else None
    x = _t(feats)
    xp = D.rows_convert(x, True)
    r = None if res is None else _t(res)
    rp = None if res is None else D.rows_convert(r, True)

    def run(mode, xin, resid, io):
        got, _, _ = D.sparse_conv_layer(xin, _t(coords), batch, shape, _t(w), (3, 3, 3), (1, 1, 1), (0, 0, 0), True,
                                        scale=_t(scale), shift=_t(shift), residual=resid, relu=True, mfma=mode, io=io)
        return got
    combos = [(x, r, 0), (xp, r, D.IO_IN_PAIR), (x, r, D.IO_OUT_PAIR), (xp, r, D.IO_IN_PAIR | D.IO_OUT_PAIR)]
    if res is not None:
        combos += [(x, rp, D.IO_RES_PAIR), (xp, rp, D.IO_IN_PAIR | D.IO_OUT_PAIR | D.IO_RES_PAIR), (xp, None, D.IO_IN_PAIR | D.IO_OUT_PAIR)]
    for xin, resid, io in combos:
        a = run("wave2_f16x3_tiles", xin, resid, io)
        b = run("r16_f16x3", xin, resid, io)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), io


def test_encoder_raster_path_equals_appearance_order_path():
    """The whole middle encoder with its level-0 rows renumbered in raster order gives, cell for cell, the bits of the
    encoder run in the voxelizer's first-appearance order (AL3D_L0=off: round 3's kernels), with the item-stream kernel
    on the 16 -> 16 layers only and on the strided 16 -> 32 layer too -- when the level-0 rows stay f32 rows
    (AL3D_L0_ROWS=f32, the default); as pair rows the outputs agree to the pair format's rounding."""
    from al3d import detector_ops as D, synthetic
    from al3d.models.backbones import FPNSpMiddleResNetFHD
    if D.MATH != "f16x3":
        pytest.skip("the raster path is the f16x3 encoder's")
    rng = np.random.default_rng(11)
    batch, in_shape = 2, [96, 88, 40]                                       # (x, y, z) as the reference passes it
    shape = [in_shape[2] + 1, in_shape[1], in_shape[0]]
    feats, coords = random_sparse(rng, batch, shape, 40001, 5)
    enc = FPNSpMiddleResNetFHD(num_input_features=5)
    synthetic.seeded_init_(enc, seed=3)
    enc = enc.to(DEV).eval()
    saved = (D.L0, set(D.R16_COUTS), D.L0_ROWS)
    outs = {}
    try:
        for tag, l0, couts, rows in (("off", "off", {16}, "f32"), ("raster16", "raster", {16}, "f32"),
                                     ("raster16+32", "raster", {16, 32}, "f32"), ("pair16", "raster", {16}, "pair"),
                                     ("pair16+32", "raster", {16, 32}, "pair")):
            D.L0, D.R16_COUTS, D.L0_ROWS = l0, couts, rows
            with torch.no_grad():
                dense, middle = enc(_t(feats), _t(coords), batch, in_shape)
            outs[tag] = (dense.clone(), [(m.features.clone(), m.indices.clone(), list(m.spatial_shape)) for m in middle])
    finally:
        D.L0, D.R16_COUTS, D.L0_ROWS = saved
        enc._packed_dev = None
    ref_dense, ref_mid = outs["off"]
    assert torch.isfinite(ref_dense).all() and float(ref_dense.abs().max()) > 0
    for tag in ("raster16", "raster16+32"):
        dense, mid = outs[tag]
        assert torch.equal(dense.view(torch.int32), ref_dense.view(torch.int32)), tag
        for (f, i, s), (rf, ri, rs) in zip(mid, ref_mid):
            assert s == rs and torch.equal(i, ri) and torch.equal(f.view(torch.int32), rf.view(torch.int32)), tag
    # level-0 rows as pair rows between the item-stream layers (AL3D_L0_ROWS=pair): the products are unchanged, the stored
    # activations carry 22-23 significant bits (what the residual adds and the next layer's split see): same outputs to
    # ~1e-6 of the scale, not the same bits
    for tag in ("pair16", "pair16+32"):
        dense, mid = outs[tag]
        err = float((dense - ref_dense).abs().max()) / float(ref_dense.abs().max())
        assert 0.0 < err <= 2e-5, (tag, err)
        for (f, i, s), (rf, ri, rs) in zip(mid, ref_mid):
            assert s == rs and torch.equal(i, ri)
            assert float((f - rf).abs().max()) <= 2e-5 * float(rf.abs().max()), tag
