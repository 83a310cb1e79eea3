"""GPU suite: detector-side HIP kernels against the CPU oracle, independent torch fp32
references and the reference's golden vectors.  Floating point throughout: tolerances are
stated per test (fp32 accumulate in a different order than the reference)."""
import logging
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_detector_oracle import G, random_sparse, to_dense

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ---------------------------------------------------------------- sparse conv layers
@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (16, 32), (32, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("subm,k,s,p", [(True, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
                                         (False, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                         (False, (3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                         (False, (3, 1, 1), (2, 1, 1), (0, 0, 0))])
@pytest.mark.parametrize("mfma", [False, True, "bf16x6", "wave", "wave2", "wave2_f16x3"])
def test_sparse_conv_layer_vs_oracle(oracle, cin, cout, subm, k, s, p, mfma):
    from al3d import detector_ops as D
    from al3d.detector_ops import MFMA_PAIRS
    if mfma and (cin, cout) not in MFMA_PAIRS:
        pytest.skip("16-wide layers run on the VALU kernel")
    rng = np.random.default_rng(cin * 131 + cout)
    shape, batch = [11, 14, 12], 2
    feats, coords = random_sparse(rng, batch, shape, 333, cin)
    w = (rng.normal(size=(*k, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    scale = (rng.uniform(0.5, 1.5, cout)).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    fo, co, oshape = oracle.spconv(feats, coords, batch, shape, w, k, s, p, subm)
    ref = np.maximum(fo * scale + shift, 0)
    got, gco, gshape = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), k, s, p, subm,
                                           scale=_t(scale), shift=_t(shift), relu=True, mfma=mfma)
    assert gshape == oshape
    got, gco = got.cpu().numpy(), gco.cpu().numpy()
    assert len(gco) == len(co)
    # row order of a strided conv's outputs is free: compare through the dense scatter
    np.testing.assert_allclose(to_dense(got, gco, batch, oshape), to_dense(ref, co, batch, oshape),
                               rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (32, 64), (64, 64), (128, 128)])
def test_sparse_conv_pipelined_kernel_is_bit_identical(cin, cout):
    """The software-pipelined wave kernel performs the same MFMA sequence per output as the
    unpipelined one: outputs must agree bit for bit (ragged tail, residual, sparse + dense taps)."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(cin + cout)
    shape, batch = [9, 40, 37], 3
    feats, coords = random_sparse(rng, batch, shape, 4001, cin)
    w = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32) if cin == cout else None
    outs = []
    for mode in ("wave", "wave2", "bf16x6"):
        got, _, _ = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), (3, 3, 3), (1, 1, 1),
                                        (0, 0, 0), True, residual=None if res is None else _t(res),
                                        relu=True, mfma=mode)
        outs.append(got.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("geom", ["subm", "down", "down311", "tiny"])
def test_sparse_conv_glds_kernel_is_bit_identical(cin, cout, geom):
    """The LDS-DMA gather kernel (csrc/spconv_glds.hip: full-line row fetches into a swizzled LDS image,
    producer-wave weight slabs) performs sp_conv_wave2's MFMA sequence per output: same bits.  Covers a
    ragged last tile, residual + ReLU, sparse and dense taps, all seven channel pairs, the encoder's strided
    geometries, and inputs smaller than one tile."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(cin * 7 + cout)
    if geom == "tiny":
        shape, batch, n = [3, 5, 4], 1, 17
    else:
        shape, batch, n = [9, 40, 37], 3, 4001
    feats, coords = random_sparse(rng, batch, shape, n, cin)
    feats[::7] *= 1e-3                                   # f16 subnormal range of the hi part / lifted residuals
    feats[5::11] *= 300.0
    k, s, p, subm = {"subm": ((3, 3, 3), (1, 1, 1), (0, 0, 0), True), "tiny": ((3, 3, 3), (1, 1, 1), (0, 0, 0), True),
                     "down": ((3, 3, 3), (2, 2, 2), (1, 1, 1), False),
                     "down311": ((3, 1, 1), (2, 1, 1), (0, 0, 0), False)}[geom]
    w = (rng.normal(size=(*k, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32) if (cin == cout and subm) else None
    outs = []
    for mode in ("wave2_f16x3", "glds_f16x3", "wave2_f16x3_tiles"):
        got, gco, _ = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), k, s, p, subm,
                                          scale=_t(scale), shift=_t(shift),
                                          residual=None if res is None else _t(res), relu=True, mfma=mode)
        outs.append((got.cpu().numpy(), gco.cpu().numpy()))
    assert outs[0][0].shape[0] > 0 and np.isfinite(outs[0][0]).all()
    # sparse_conv_layer claims strided output sites with atomics (row order differs from run to run):
    # compare site by site through the dense scatter
    _, _, oshape = D.sparse_conv_layer(_t(feats[:1]), _t(coords[:1]), batch, shape, _t(w), k, s, p, subm, mfma=False)
    a = to_dense(outs[0][0], outs[0][1], batch, oshape)
    b = to_dense(outs[1][0], outs[1][1], batch, oshape)
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    c = to_dense(outs[2][0], outs[2][1], batch, oshape)          # the wave kernel reading the tiled rulebook's masks
    assert np.array_equal(a.view(np.int32), c.view(np.int32))


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64)])
@pytest.mark.parametrize("order", ["raster", "random", "tiny"])
def test_sparse_conv_range_gather_kernel(cin, cout, order):
    """The range-gather LDS-DMA kernel (csrc/spconv_rng.hip): one staged index range per (tile, kz, ky) serves the
    three kx taps.  Rows in raster order (the encoder's levels 1+): short ranges, the staged path; rows in random
    order: every range is long, the per-row fallback inside the same kernel.  Cin = 32: the MFMA sequence per
    output is sp_conv_wave2's -- same bits.  Cin = 64: the two 32-channel chunks of a group are summed before the
    next group (another summation order): agreement to 2e-6 of the output scale, and run-to-run identical."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(cin + len(order))
    if order == "tiny":
        shape, batch, n = [3, 5, 4], 1, 17
    else:
        shape, batch, n = [9, 40, 37], 3, 6001
    feats, coords = random_sparse(rng, batch, shape, n, cin)
    if order != "random":
        key = ((coords[:, 0].astype(np.int64) * shape[0] + coords[:, 1]) * shape[1] + coords[:, 2]) * shape[2] + coords[:, 3]
        perm = np.argsort(key, kind="stable")
        feats, coords = feats[perm], coords[perm]
    feats[::7] *= 1e-3
    feats[5::11] *= 300.0
    w = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32)
    outs = {}
    for mode in ("wave2_f16x3", "rng_f16x3", "rng_f16x3 again"):
        got, _, _ = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), (3, 3, 3), (1, 1, 1), (0, 0, 0), True,
                                        scale=_t(scale), shift=_t(shift), residual=_t(res), relu=True,
                                        mfma=mode.split()[0])
        outs[mode] = got.cpu().numpy()
    ref, got = outs["wave2_f16x3"], outs["rng_f16x3"]
    assert np.isfinite(ref).all() and ref.shape == got.shape
    assert np.array_equal(got.view(np.int32), outs["rng_f16x3 again"].view(np.int32))
    if cin == 32:
        assert np.array_equal(ref.view(np.int32), got.view(np.int32))
    else:
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("mode,cin,cout", [("wave2_f16x3_tiles", 16, 16), ("wave2_f16x3_tiles", 16, 32),
                                           ("wave2_f16x3_tiles", 64, 128), ("wave2_f16x3_tiles", 128, 128),
                                           ("glds_f16x3", 32, 32), ("glds_f16x3", 64, 64), ("rng_f16x3", 32, 32),
                                           ("rng_f16x3", 64, 64)])
def test_sparse_conv_pair_rows(mode, cin, cout):
    """Pair rows (csrc/sp_rows.h): activations stored as the two f16 planes of the f16x3 arithmetic, split once
    in the producer's epilogue.  (1) a layer fed pair rows gives the BITS of the layer fed the f32 rows they were
    split from (the kernels only ever multiplied xh and xl'); (2) writing pair rows = splitting the f32 output;
    (3) a pair-row residual adds xh + xl' 2^-11; (4) all three together compose.  Round trip error of the
    format: <= 2^-22 relative (2^-36 absolute below f16's normal range)."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(cin * 3 + cout)
    shape, batch, n = [9, 40, 37], 3, 5003
    feats, coords = random_sparse(rng, batch, shape, n, cin)
    key = ((coords[:, 0].astype(np.int64) * shape[0] + coords[:, 1]) * shape[1] + coords[:, 2]) * shape[2] + coords[:, 3]
    perm = np.argsort(key, kind="stable")
    feats, coords = feats[perm], coords[perm]
    feats[::7] *= 1e-3
    feats[5::11] *= 300.0
    w = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    res = rng.normal(size=(feats.shape[0], cout)).astype(np.float32)
    x, r = _t(feats), _t(res)
    xp, rp = D.rows_convert(x, True), D.rows_convert(r, True)
    back = D.rows_convert(xp, False)
    # 2^-22 relative in f16's normal range, 2^-36 absolute below it (hi piece subnormal: conv2d_f16x3.hip header)
    assert bool(((back - x).abs() <= torch.maximum(x.abs() * 2.0 ** -22, torch.tensor(2.0 ** -35, device=x.device))).all())
    r_rounded = D.rows_convert(rp, False)

    def run(xin, resid, io):
        got, _, _ = D.sparse_conv_layer(xin, _t(coords), batch, shape, _t(w), (3, 3, 3), (1, 1, 1), (0, 0, 0), True,
                                        scale=_t(scale), shift=_t(shift), residual=resid, relu=True, mfma=mode, io=io)
        return got

    def same(a, b):
        return bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))
    base = run(x, r, 0)
    assert same(run(xp, r, D.IO_IN_PAIR), base)                                           # (1)
    assert same(D.rows_convert(run(x, r, D.IO_OUT_PAIR), False),
                D.rows_convert(D.rows_convert(base, True), False))                        # (2)
    base_rr = run(x, r_rounded, 0)
    assert same(run(x, rp, D.IO_RES_PAIR), base_rr)                                       # (3)
    allp = run(xp, rp, D.IO_IN_PAIR | D.IO_OUT_PAIR | D.IO_RES_PAIR)
    assert same(allp, D.rows_convert(base_rr, True))                                      # (4)
    assert same(run(xp, None, D.IO_IN_PAIR | D.IO_OUT_PAIR), D.rows_convert(run(x, None, 0), True))


@pytest.mark.parametrize("k,s,p,subm", [((1, 1, 3), (1, 1, 2), (0, 0, 0), False),     # BEVFusion-style conv_out
                                        ((3, 3, 1), (1, 1, 1), (0, 0, 0), True),
                                        ((1, 3, 3), (1, 2, 2), (0, 1, 1), False),
                                        ((3, 3, 3), (1, 1, 1), (0, 0, 0), True),
                                        ((2, 2, 2), (2, 2, 2), (0, 0, 0), False)])
def test_sparse_conv_anisotropic_geometries(oracle, k, s, p, subm):
    """Kernel/stride/padding combinations beyond the ones FPNSpMiddleResNetFHD uses (the rulebook and
    the conv kernels are generic in them), 32 -> 64 channels on the default matrix-core path."""
    from al3d import detector_ops as D
    rng = np.random.default_rng(sum(k) * 7 + sum(s))
    shape, batch = [7, 19, 23], 2
    feats, coords = random_sparse(rng, batch, shape, 700, 32)
    w = (rng.normal(size=(*k, 32, 64)) / np.sqrt(32 * np.prod(k))).astype(np.float32)
    fo, co, oshape = oracle.spconv(feats, coords, batch, shape, w, k, s, p, subm)
    got, gco, gshape = D.sparse_conv_layer(_t(feats), _t(coords), batch, shape, _t(w), k, s, p, subm, relu=False)
    assert gshape == oshape and len(gco) == len(co)
    np.testing.assert_allclose(to_dense(got.cpu().numpy(), gco.cpu().numpy(), batch, oshape),
                               to_dense(fo, co, batch, oshape), rtol=2e-5, atol=2e-5)


def test_sparse_conv_residual_and_empty(oracle):
    from al3d import detector_ops as D
    rng = np.random.default_rng(2)
    feats, coords = random_sparse(rng, 1, [5, 6, 7], 40, 16)
    w = rng.normal(size=(3, 3, 3, 16, 16)).astype(np.float32) * 0.1
    res = rng.normal(size=(40, 16)).astype(np.float32)
    fo, _, _ = oracle.spconv(feats, coords, 1, [5, 6, 7], w, (3, 3, 3), (1, 1, 1), (0, 0, 0), True)
    got, _, _ = D.sparse_conv_layer(_t(feats), _t(coords), 1, [5, 6, 7], _t(w), (3, 3, 3), (1, 1, 1),
                                    (0, 0, 0), True, residual=_t(res), relu=True)
    np.testing.assert_allclose(got.cpu().numpy(), np.maximum(fo + res, 0), rtol=2e-5, atol=2e-5)
    e, ec, _ = D.sparse_conv_layer(torch.zeros((0, 16), device=DEV), torch.zeros((0, 4), dtype=torch.int32,
                                   device=DEV), 1, [5, 6, 7], _t(w), (3, 3, 3), (2, 2, 2), (1, 1, 1), False)
    assert e.shape[0] == 0 and ec.shape[0] == 0


# ---------------------------------------------------------------- the encoder's OWN index path vs the oracle
def _one_stage_encoder(mods):
    """A one-stage encoder built from the product's `_SparseEncoderBase`: its `_run` goes through
    `build_rulebook` (al3d_sp_scatter_index / al3d_sp_subm_table / al3d_sp_down_sites /
    al3d_sp_down_table), `_pack` (incl. the zero-padded 5 -> 16 first layer) and `_conv` -- the code the
    shipped FPNSpMiddleResNetFHD runs, unlike `detector_ops.sparse_conv_layer` (al3d_sp_down_claim)."""
    from torch import nn
    from al3d.models.backbones import _SparseEncoderBase

    class OneStage(_SparseEncoderBase):
        def __init__(self):
            super().__init__()
            self.stage = nn.Sequential(*mods)

        def _stages(self):
            return [self.stage]
    return OneStage()


def _np_fold(bn):
    inv = 1.0 / np.sqrt(bn.running_var.numpy().astype(np.float64) + bn.eps)
    scale = bn.weight.detach().numpy() * inv
    return scale.astype(np.float32), (bn.bias.detach().numpy() - bn.running_mean.numpy() * scale).astype(np.float32)


@pytest.mark.parametrize("math", ["f16x3", "bf16x6", "f32"])
@pytest.mark.parametrize("cin,cout,subm,k,s,p", [
    (5, 16, True, (3, 3, 3), (1, 1, 1), (0, 0, 0)),          # first layer: zero-padded to 16 input channels
    (16, 32, False, (3, 3, 3), (2, 2, 2), (1, 1, 1)),        # the three strided geometries of the encoder
    (64, 128, False, (3, 3, 3), (2, 2, 2), (0, 1, 1)),
    (128, 128, False, (3, 1, 1), (2, 1, 1), (0, 0, 0)),
    (32, 32, True, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    (64, 64, True, (3, 3, 3), (1, 1, 1), (0, 0, 0))])
def test_encoder_rulebook_path_single_layer_vs_oracle(oracle, math, cin, cout, subm, k, s, p):
    """scn.py:331-369 layer by layer through the product's rulebook + packing + conv dispatch."""
    from al3d import detector_ops as D, synthetic
    from al3d.models.backbones import SparseConv3d, SubMConv3d, _bn
    from torch import nn
    rng = np.random.default_rng(cin * 17 + cout + sum(s))
    shape, batch = [9, 40, 37], 3
    feats, coords = random_sparse(rng, batch, shape, 4001, cin)
    conv = SubMConv3d(cin, cout, k, bias=False) if subm else SparseConv3d(cin, cout, k, s, padding=p, bias=False)
    enc = _one_stage_encoder([conv, _bn(cout), nn.ReLU()])
    synthetic.seeded_init_(enc, seed=cin + cout)
    enc.eval()
    sc, sh = _np_fold(enc.stage[1])
    fo, co, oshape = oracle.spconv(feats, coords, batch, shape, conv.weight.detach().numpy(), k, s, p, subm)
    ref = np.maximum(fo * sc + sh, 0)
    saved = D.MATH
    try:
        D.MATH = math
        enc = enc.to(DEV)
        with torch.no_grad():
            final, middle = enc._run(_t(feats), _t(coords), batch, shape)
    finally:
        D.MATH = saved
    assert list(final.spatial_shape) == oshape and final.features.shape == (len(co), cout)
    got, gco = final.features.cpu().numpy(), final.indices.cpu().numpy()
    if subm:
        assert np.array_equal(gco, coords)              # SubM keeps the sites and their order
    np.testing.assert_allclose(to_dense(got, gco, batch, oshape), to_dense(ref, co, batch, oshape),
                               rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("c", [16, 128])
def test_encoder_rulebook_path_basic_block_vs_oracle(oracle, c):
    """SparseBasicBlock (scn.py:54-97): conv+bias, BN, ReLU, conv+bias, BN, +identity, ReLU through the
    product path; both convs share one rulebook (indice_key)."""
    from al3d import synthetic
    from al3d.models.backbones import SparseBasicBlock
    rng = np.random.default_rng(c)
    shape, batch = [9, 40, 37], 2
    feats, coords = random_sparse(rng, batch, shape, 3000, c)
    blk = SparseBasicBlock(c, c, indice_key="res")
    enc = _one_stage_encoder([blk])
    synthetic.seeded_init_(enc, seed=c)
    enc.eval()
    k3, s1, p0 = (3, 3, 3), (1, 1, 1), (0, 0, 0)
    sc1, sh1 = _np_fold(blk.bn1)
    sc2, sh2 = _np_fold(blk.bn2)
    f, _, _ = oracle.spconv(feats, coords, batch, shape, blk.conv1.weight.detach().numpy(), k3, s1, p0, True)
    f = np.maximum((f + blk.conv1.bias.detach().numpy()) * sc1 + sh1, 0)
    f, _, _ = oracle.spconv(f, coords, batch, shape, blk.conv2.weight.detach().numpy(), k3, s1, p0, True)
    ref = np.maximum((f + blk.conv2.bias.detach().numpy()) * sc2 + sh2 + feats, 0)
    enc = enc.to(DEV)
    with torch.no_grad():
        final, _ = enc._run(_t(feats), _t(coords), batch, shape)
    np.testing.assert_allclose(final.features.cpu().numpy(), ref, rtol=5e-5, atol=5e-5)


# ---------------------------------------------------------------- full encoder vs dense torch
def _dense_encoder_reference(model, feats, coords, batch, shape):
    """FPNSpMiddleResNetFHD emulated with dense conv3d + site masks on CPU (independent of the
    rulebook code): SubM = conv3d(pad 1) masked to the active sites; SparseConv3d = strided
    conv3d, active sites = max-pooled mask; BN(eval)/ReLU applied on active sites only."""
    x = torch.from_numpy(to_dense(feats, coords, batch, shape))
    mask = torch.from_numpy(to_dense(np.ones((len(coords), 1), np.float32), coords, batch, shape))

    def conv(mod, x):
        w = mod.weight.detach().permute(4, 3, 0, 1, 2).contiguous()
        pad = [q // 2 for q in mod.kernel_size] if mod.subm else list(mod.padding)
        y = F.conv3d(x, w, stride=mod.stride, padding=pad)
        if mod.bias is not None:
            y = y + mod.bias.detach().view(1, -1, 1, 1, 1)
        return y

    def bn(mod, y):
        sh = (1, -1, 1, 1, 1)
        return (y - mod.running_mean.view(sh)) / torch.sqrt(mod.running_var.view(sh) + mod.eps) \
            * mod.weight.detach().view(sh) + mod.bias.detach().view(sh)

    for seq in model._stages():
        mods = list(seq.children())
        i = 0
        while i < len(mods):
            m = mods[i]
            name = type(m).__name__
            if name in ("SubMConv3d", "SparseConv3d"):
                if not m.subm:
                    mask = (F.max_pool3d(mask, m.kernel_size, m.stride, m.padding) > 0).float()
                x = torch.relu(bn(mods[i + 1], conv(m, x))) * mask
                i += 3
            elif name == "SparseBasicBlock":
                idt = x
                y = torch.relu(bn(m.bn1, conv(m.conv1, x))) * mask
                y = bn(m.bn2, conv(m.conv2, y)) * mask
                x = torch.relu(y + idt) * mask
                i += 1
            else:
                i += 1
    B, C, D_, H_, W_ = x.shape
    return x.reshape(B, C * D_, H_, W_).permute(0, 2, 3, 1).contiguous().numpy()   # NHWC, ch = c*D+z


def test_sparse_encoder_vs_dense_torch():
    from al3d import synthetic
    from al3d.models.backbones import FPNSpMiddleResNetFHD
    rng = np.random.default_rng(4)
    grid_xyz = [32, 32, 40]
    shape = [41, 32, 32]
    batch = 2
    feats, coords = random_sparse(rng, batch, [40, 32, 32], 2500, 5)
    enc = FPNSpMiddleResNetFHD(num_input_features=5)
    synthetic.seeded_init_(enc, seed=3)
    enc.eval()
    ref = _dense_encoder_reference(enc, feats, coords, batch, shape)
    enc = enc.to(DEV)
    with torch.no_grad():
        out, middle = enc(_t(feats), _t(coords), batch, np.array(grid_xyz))
        out2, _ = enc(_t(feats), _t(coords), batch, np.array(grid_xyz))   # level grids left clean?
    assert out.shape == (batch, 4, 4, 256) and len(middle) == 4
    assert torch.equal(out, out2)                                         # deterministic, bitwise
    # 21 stacked fp32 convs; activations are O(1): absolute tolerance dominates
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-3, atol=2e-4)


# ---------------------------------------------------------------- dense neck vs the reference RPN
def test_rpn_matches_reference_golden():
    from al3d import synthetic
    from al3d.models.necks import RPN
    z = np.load(os.path.join(G, "rpn.npz"))
    net = RPN(layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
              us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256,
              norm_cfg=None, logger=logging.getLogger("RPN"))
    synthetic.seeded_init_(net, seed=1234)
    import hashlib
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    digest = hashlib.sha256(np.concatenate([sd[k].ravel().astype(np.float64) for k in sorted(sd)]).tobytes())
    assert digest.hexdigest() == str(z["state_sha256"]), "weights differ from the golden run"
    net = net.to(DEV).eval()
    x = torch.from_numpy(z["x"].astype(np.float32)).permute(0, 2, 3, 1).contiguous().to(DEV)
    with torch.no_grad():
        y = net(x)                                         # NHWC [1,32,48,512]
    from al3d import detector_ops as D
    emb = D.gap_nhwc(y).cpu().numpy()
    # the embedding the deblock launches emit themselves (fused GAP: workgroup partial sums, ascending order) against
    # the stand-alone W-then-H kernel on the stored map: same numbers up to the summation order
    if D.MATH == "f16x3" and D.GAP == "fused":
        assert net.embedding is not None and net.embedding.shape == (1, 512)
        np.testing.assert_allclose(net.embedding.cpu().numpy(), emb, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(net.embedding.cpu().numpy(), z["emb"], rtol=1e-4, atol=1e-5)
    y = y.permute(0, 3, 1, 2).cpu().numpy()
    # 7 stacked 3x3 fp32 convs with BN: values O(1); reference ran on CPU torch
    np.testing.assert_allclose(y[0, ::16, ::4, ::4], z["y_slice"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(emb, z["emb"], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- head: decode, fused convs, NMS
def test_box_decode_matches_reference_golden(oracle):
    from al3d import detector_ops as D
    z = np.load(os.path.join(G, "decode.npz"))
    got = D.box_decode(_t(z["enc"]), _t(z["anchors"])).cpu().numpy()
    ref = z["dec"].reshape(-1, 9)
    np.testing.assert_allclose(got[:, :8], ref[:, :8], rtol=2e-6, atol=2e-6)
    d = np.abs(got[:, 8] - ref[:, 8])
    assert np.minimum(d, 2 * np.pi - d).max() < 1e-5
    orc = oracle.box_decode(z["enc"].reshape(-1, 10), z["anchors"].reshape(-1, 9))
    np.testing.assert_allclose(got, orc, rtol=2e-6, atol=2e-6)


def _head_setup(seed, H=16, W=16, score_bias=0.0):
    from al3d import synthetic
    from al3d.datasets.anchors import generate_task_anchors
    from al3d.models import build_head
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal.py"))
    head = build_head(cfg.model.bbox_head)
    synthetic.seeded_init_(head, seed=seed)
    with torch.no_grad():
        for t in head.tasks:
            t.conv_cls.bias.add_(score_bias)
    # anchors of a small BEV map: same generator, smaller feature map
    gens = [dict(g, anchor_ranges=[-12.8, -12.8, g["anchor_ranges"][2], 12.8, 12.8, g["anchor_ranges"][5]])
            for g in cfg.target_assigner.anchor_generators]
    anchors = generate_task_anchors(cfg.tasks, gens, [1, H, W])
    return cfg, head.to(DEV).eval(), anchors


@pytest.mark.parametrize("score_bias", [0.0, -4.0])
def test_head_predict_vs_oracle(oracle, score_bias):
    """score_bias 0: nearly every anchor passes the 0.1 threshold (top-k of 1000 and heavy
    suppression are exercised); -4: few candidates, some tasks empty."""
    B, H, W = 2, 16, 16
    cfg, head, anchors = _head_setup(7, H, W, score_bias)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, 512, generator=g).to(DEV)
    tc = cfg.test_cfg
    with torch.no_grad():
        preds = head(x)
        fused = preds[0]["_fused"]
        # fused 1x1 convs == the twelve separate Conv2d of the reference head
        xn = x.permute(0, 3, 1, 2).cpu()
        for t, task in enumerate(head.tasks):
            ref = F.conv2d(xn, task.conv_box.weight.cpu(), task.conv_box.bias.cpu()).permute(0, 2, 3, 1)
            torch.testing.assert_close(preds[t]["box_preds"].cpu(), ref, rtol=1e-4, atol=1e-4)
        out = head.predict({"anchors": [_t(a) for a in anchors], "metadata": [{"i": i} for i in range(B)]},
                           preds, tc)
    fused = fused.cpu().numpy().reshape(B, H * W, -1)
    label_off = np.concatenate([[0], np.cumsum(head.num_classes)])
    for b in range(B):
        bb, ss, ll = [], [], []
        for t in range(len(head.tasks)):
            bx, sc, lb = oracle.head_predict(fused[b], anchors[t], head.num_anchor_per_locs[t],
                                             head.num_classes[t], head._box_off[t], head._cls_off[t],
                                             tc.score_threshold, tc.nms.nms_iou_threshold,
                                             tc.nms.nms_pre_max_size, tc.nms.nms_post_max_size,
                                             tc.post_center_limit_range)
            bb.append(bx); ss.append(sc); ll.append(lb + label_off[t])
        bb, ss, ll = np.concatenate(bb), np.concatenate(ss), np.concatenate(ll)
        got = out[b]
        assert got["metadata"] == {"i": b}
        assert got["label_preds"].dtype == torch.int64
        # discrete outcome (which boxes survive, in which order) must agree exactly
        assert got["label_preds"].cpu().numpy().tolist() == ll.tolist()
        np.testing.assert_allclose(got["scores"].cpu().numpy(), ss, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got["box3d_lidar"].cpu().numpy()[:, :8], bb[:, :8], rtol=1e-4, atol=1e-4)
        if score_bias == 0.0:
            assert len(ss) > 50


def test_head_predict_matches_the_reference_predict_golden():
    """The product's ``MultiGroupHead.predict`` (decode + NMS kernel over the fused head output) against the REFERENCE's own
    ``MultiGroupHead.predict`` (mg_head.py:697-803; oracle/gen_golden_head_predict.py): six tasks, two samples, one (sample,
    task) without a candidate.  Same detections in the same order; scores 1e-6; boxes 1e-4 (device exp / atan2)."""
    z = np.load(os.path.join(G, "head_predict_tasks.npz"))
    B, H, W = (int(v) for v in z["shape"])
    cfg, head, anchors = _head_setup(7, H, W)
    assert head.num_classes == [int(v) for v in z["num_classes"]]
    head._prepare(torch.device(DEV))
    fused = torch.zeros((B, H, W, head._ch), dtype=torch.float32)
    for t, task in enumerate(head.tasks):
        np.testing.assert_array_equal(anchors[t], z[f"anchors{t}"])
        fused[..., head._box_off[t]:head._box_off[t] + task.conv_box.out_channels] = torch.from_numpy(z[f"box{t}"])
        fused[..., head._cls_off[t]:head._cls_off[t] + task.conv_cls.out_channels] = torch.from_numpy(z[f"cls{t}"])
    fused = fused.to(DEV)
    out = head.predict({"anchors": [_t(a) for a in anchors], "metadata": [None] * B},
                       [{"_fused": fused} for _ in head.tasks], cfg.test_cfg)
    for b in range(B):
        rb, rs, rl = z[f"out{b}.boxes"], z[f"out{b}.scores"], z[f"out{b}.labels"]
        got = out[b]
        assert got["label_preds"].cpu().numpy().tolist() == rl.tolist()
        np.testing.assert_allclose(got["scores"].cpu().numpy(), rs, rtol=1e-6, atol=1e-7)
        gb = got["box3d_lidar"].cpu().numpy()
        np.testing.assert_allclose(gb[:, :8], rb[:, :8], rtol=1e-4, atol=1e-4)
        d = np.abs(gb[:, 8] - rb[:, 8])
        assert np.minimum(d, 2 * np.pi - d).max() < 1e-4


@pytest.mark.parametrize("levels", [3, 40])
def test_head_predict_with_massive_score_ties(oracle, levels):
    """Class logits quantised to a few levels: thousands of anchors share the score of the 1000th candidate, so the
    top-k cut falls inside a run of exact ties (lower anchor index first).  levels=3: far more ties than
    the kernel's 1024-entry tie list (chunked fallback); levels=40: tens to hundreds (list path)."""
    B, H, W = 2, 16, 16
    cfg, head, anchors = _head_setup(11, H, W, 0.0)
    head._prepare(torch.device(DEV))
    g = torch.Generator().manual_seed(levels)
    fused = torch.randn(B, H, W, head._ch, generator=g)
    for t, task in enumerate(head.tasks):
        c0 = head._cls_off[t]
        cls = fused[..., c0:c0 + task.conv_cls.out_channels]
        fused[..., c0:c0 + task.conv_cls.out_channels] = torch.round(cls * (levels / 4.0)) / (levels / 4.0)
    fused = fused.to(DEV).contiguous()
    preds = [{"_fused": fused} for _ in head.tasks]
    tc = cfg.test_cfg
    out = head.predict({"anchors": [_t(a) for a in anchors], "metadata": [{"i": i} for i in range(B)]}, preds, tc)
    fnp = fused.cpu().numpy().reshape(B, H * W, -1)
    label_off = np.concatenate([[0], np.cumsum(head.num_classes)])
    for b in range(B):
        bb, ss, ll = [], [], []
        for t in range(len(head.tasks)):
            bx, sc, lb = oracle.head_predict(fnp[b], anchors[t], head.num_anchor_per_locs[t], head.num_classes[t],
                                             head._box_off[t], head._cls_off[t], tc.score_threshold,
                                             tc.nms.nms_iou_threshold, tc.nms.nms_pre_max_size,
                                             tc.nms.nms_post_max_size, tc.post_center_limit_range)
            bb.append(bx); ss.append(sc); ll.append(lb + label_off[t])
        bb, ss, ll = np.concatenate(bb), np.concatenate(ss), np.concatenate(ll)
        got = out[b]
        assert got["label_preds"].cpu().numpy().tolist() == ll.tolist()
        np.testing.assert_allclose(got["scores"].cpu().numpy(), ss, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got["box3d_lidar"].cpu().numpy()[:, :8], bb[:, :8], rtol=1e-4, atol=1e-4)
        assert len(ss) > 20


# ---------------------------------------------------------------- detector end to end
def test_detector_batch_invariance_and_determinism():
    from al3d import synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.sweep import sweep_embeddings
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_numpy([synthetic.make_point_cloud(30 + i, nsweeps=3) for i in range(3)], DEV)
    e1 = sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 1, device=DEV), DEV, 3)
    e3 = sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 3, device=DEV), DEV, 3)
    e3b = sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 3, device=DEV), DEV, 3)
    assert e1.shape == (3, 512) and torch.isfinite(e1).all()
    assert torch.equal(e3, e3b)                      # run-to-run bitwise reproducible
    # batching only changes row numbering of the sparse tensors, not any per-row arithmetic
    assert torch.equal(e1, e3)
    ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 3, device=DEV)))
    with torch.no_grad():
        preds, middle = model(ex, return_loss=False, estimate=True)
    assert len(preds) == 3 and tuple(middle[-1].shape) == (3, 512, 128, 128)
    for p in preds:
        k = p["box3d_lidar"].shape[0]
        assert p["box3d_lidar"].shape == (k, 9) and p["scores"].shape == (k,) and k <= 6 * 83
        assert p["label_preds"].dtype == torch.int64 and int(p["label_preds"].max()) <= 9
        assert p["metadata"]["token"].startswith("frame")


def test_full_size_frames_kernel_structures_and_pipeline_agree():
    """BASELINE-size inputs (10-sweep ~250k-point frames, 60k-voxel cap): properties that need no
    oracle at this size -- the three sparse-conv kernel structures give bit-identical embeddings,
    the two-stream batch pipeline gives the same bits as the serial sweep, and the sweep is
    invariant to the batch size."""
    from al3d import detector_ops as D, sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_synthetic(6, DEV, num_base=3, seed=7)
    assert min(f.shape[0] for f in pool.frames) > 200000

    def run(batch):
        return S.sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch, device=DEV),
                                  DEV, len(pool))
    saved = (D.SPCONV, S.PIPELINE)
    saved_dense = D.DENSE
    try:
        ref = run(4)
        assert ref.shape == (6, 512) and torch.isfinite(ref).all()
        assert torch.equal(run(3), ref)
        # bf16x6: "wave" / "tile" structures; f16x3: the LDS-DMA gather kernel everywhere ("glds"), nowhere
        # ("wave2"), or per channel pair (the default) -- one arithmetic each, same bits
        # (round 4: the default also renumbers the level-0 rows in raster order and runs their layers as item streams,
        # csrc/spconv_l0.hip; "glds" / "wave2" keep the voxelizer's row order.  Same bits as long as the level-0 rows travel
        # as f32 rows -- the default -- in all of them; as pair rows the embedding moves by the pair format's rounding)
        saved_l0 = D.L0_ROWS
        try:
            D.L0_ROWS = "f32"
            ref_f32 = run(4) if D.MATH == "f16x3" else ref
            for mode in (("wave", "tile") if D.MATH == "bf16x6" else ("glds", "wave2")):
                D.SPCONV = mode
                assert torch.equal(run(4), ref_f32), mode
            D.SPCONV = saved[0]
            if D.MATH == "f16x3" and D.sparse_raster():
                D.L0_ROWS = "pair"
                ref_pair = run(4)
                assert float((ref_f32 - ref_pair).abs().max()) <= 2e-5 * float(ref_f32.abs().max())
                assert not torch.equal(ref_f32, ref_pair)   # the knob does something
        finally:
            D.L0_ROWS = saved_l0
        if saved_l0 == "f32":
            assert torch.equal(ref, ref_f32)                # the default IS the bit-compatible form
        # f16x3 dense structures: LDS-staged kernels everywhere / round 1's streamed-weight policy, against the
        # default (3x3 streamed fragments + LDS-DMA kernel for the other geometries, fused GAP in all of them)
        for mode in ("lds", "stream"):
            D.DENSE = mode
            assert torch.equal(run(4), ref), "dense " + mode
        D.DENSE = saved_dense
        if D.MATH == "f16x3":
            # row format between sparse layers: pair rows (default) store xh + xl' 2^-11 instead of the f32 value;
            # the products are the same, the residual adds see the rounded value -> same embedding to ~1e-6
            saved_rows = D.SPROWS
            try:
                D.SPROWS = "f32"
                plain = run(4)
            finally:
                D.SPROWS = saved_rows
            assert float((plain - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
            assert not torch.equal(plain, ref)          # the knob does something
            # pair pixels inside the neck (block outputs -> stride-2 conv / deblocks, concat map -> head): products
            # unchanged, the fused GAP sums the f32 values in both cases -> the SAME embedding bits
            saved_pix = D.DPIX
            try:
                D.DPIX = "f32"
                assert torch.equal(run(4), ref), "dense pair pixels changed the embedding"
            finally:
                D.DPIX = saved_pix
        for mode in (None, "ahead", "split"):
            S.PIPELINE = mode
            assert torch.equal(run(2), ref), mode
        # placement knobs of the "ahead" pipeline (side work released beside the dense neck, decode + NMS launch held
        # back until the next batch's sparse encoder is through, convolutions on a high-priority stream): same bits
        S.PIPELINE = "ahead"
        knobs = (S.SIDE_AFTER_SPARSE, S.NMS_AFTER_SPARSE, S.MAIN_PRIORITY)
        try:
            for flags in ((True, False, False), (True, True, False), (False, False, True)):
                S.SIDE_AFTER_SPARSE, S.NMS_AFTER_SPARSE, S.MAIN_PRIORITY = flags
                assert torch.equal(run(2), ref), flags
            assert not getattr(model.bbox_head, "defer_nms", False) and not getattr(model.bbox_head, "_deferred", [])
        finally:
            S.SIDE_AFTER_SPARSE, S.NMS_AFTER_SPARSE, S.MAIN_PRIORITY = knobs
        # a deferred decode + NMS launch yields the detections of an immediate one, whether it is flushed or read first
        ex = next(iter(DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 2, device=DEV)))
        with torch.no_grad():
            now, _ = model(ex, return_loss=False, estimate=True)
            model.bbox_head.defer_nms = True
            try:
                later, _ = model(ex, return_loss=False, estimate=True)
                lazy, _ = model(ex, return_loss=False, estimate=True)
                assert len(model.bbox_head._deferred) == 2
                model.bbox_head.flush_deferred()
            finally:
                model.bbox_head.defer_nms = False
        for b in range(2):
            for k in ("box3d_lidar", "scores", "label_preds"):
                assert torch.equal(later[b][k], now[b][k]) and torch.equal(lazy[b][k], now[b][k]), k
    finally:
        D.SPCONV, S.PIPELINE = saved
        D.DENSE = saved_dense


def test_batch_128_equals_batch_8_at_full_size():
    """The bench's default batch (128 frames per launch; 64 until round 2) puts 5.5e9 cells into the level-0 index
    grid and the voxelizer's first-index grid -- beyond int32 (already at 64: 2.7e9).  Embeddings must be the same
    bits as with 8 frames per launch (64-bit cell indexing everywhere), at 64 and at 128."""
    from al3d import sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_synthetic(128, DEV, num_base=4, seed=3)

    def run(batch):
        return S.sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch, device=DEV),
                                  DEV, len(pool))
    small = run(8)
    for batch in (64, 128):
        big = run(batch)
        assert big.shape == (128, 512) and torch.isfinite(big).all()
        assert torch.equal(big, small), batch
    assert not torch.equal(small[0], small[1])      # frames differ (the last frame is not a copy of the first)
    del pool
    torch.cuda.empty_cache()


def test_sweep_with_an_empty_frame_in_the_batch():
    """A frame without a single point in range (ragged batch: 0 voxels for one sample) must not disturb
    its neighbours: their embeddings are the bits they have without it, the empty frame's embedding is
    that of an all-zero BEV map (finite, the same for any position in the batch), and its detections
    are whatever the head makes of the biases -- no crash, no NaN."""
    from al3d import sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    clouds = [synthetic.make_point_cloud(500 + i, nsweeps=1) for i in range(3)]
    far = np.array([[500.0, 500.0, 0.0, 1.0, 0.0]], dtype=np.float32)          # one point, outside the range
    none = np.zeros((0, 5), dtype=np.float32)

    def run(frames, batch):
        pool = PoolFrames.from_numpy(frames, DEV)
        return S.sweep_embeddings(model, DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch, device=DEV),
                                  DEV, len(pool))
    ref = run(clouds, 3)
    a = run([clouds[0], far, clouds[1], clouds[2]], 4)
    b = run([none, clouds[0], clouds[1], clouds[2], far], 5)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(a[[0, 2, 3]], ref) and torch.equal(b[[1, 2, 3]], ref)
    assert torch.equal(a[1], b[0]) and torch.equal(a[1], b[4])
    assert torch.equal(run([far], 1)[0], a[1])                                 # a batch that is empty altogether


def test_pipelined_sweep_keeps_a_bounded_number_of_batches_alive():
    """The side stream is ~10x faster than the main one; without a bound on its run-ahead it prepares
    (and keeps alive) every remaining batch of the pool.  Peak memory of a 16-batch sweep must stay
    within a small factor of the serial sweep's."""
    from al3d import sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_numpy([synthetic.make_point_cloud(200 + i, nsweeps=2) for i in range(32)], DEV)
    saved = S.PIPELINE
    peaks = {}
    try:
        for mode in (None, "ahead", "split"):
            S.PIPELINE = mode
            loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 2, device=DEV)
            S.sweep_embeddings(model, loader, DEV, len(pool))           # warm caches / level grids
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            S.sweep_embeddings(model, loader, DEV, len(pool))
            torch.cuda.synchronize()
            peaks[mode] = torch.cuda.max_memory_allocated() - base
    finally:
        S.PIPELINE = saved
    assert peaks["ahead"] < 3.0 * peaks[None] and peaks["split"] < 3.0 * peaks[None], peaks


def test_sweep_rejects_activations_beyond_the_f16x3_range():
    """f16x3 carries |activation| < 65504.  A model whose first neck BN multiplies by 1e7 must make the
    sweep fail loudly (non-finite embeddings -> Al3dError naming AL3D_MATH=bf16x6), not return numbers."""
    from al3d import detector_ops as D, sweep as S, synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.lib import Al3dError
    from al3d.models import build_detector
    from al3d.utils import Config
    if D.MATH != "f16x3":
        pytest.skip("range check applies to the f16x3 arithmetic")
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    with torch.no_grad():
        model.neck.blocks[0][2].weight.mul_(1.0e7)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    pool = PoolFrames.from_numpy([synthetic.make_point_cloud(300 + i, nsweeps=1) for i in range(2)], DEV)
    loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 2, device=DEV)
    with pytest.raises(Al3dError, match="AL3D_MATH=bf16x6"):
        S.sweep_embeddings(model, loader, DEV, len(pool))


def test_uncertainty_sweeps_compose(oracle, tmp_path):
    """pred=True paths of Entropy / Badge / UWE: the swept quantities must equal what the
    reference expressions give on the detector's own outputs (entropy from the post-NMS scores,
    embeddings times entropy, UWE's batch-local weighting)."""
    import json
    import pickle
    from al3d import synthetic
    from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
    from al3d.models import build_detector
    from al3d.selectors import build_selector
    from al3d.sweep import sweep_embeddings
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal_feature.py"))
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synthetic.seeded_init_(model, seed=0)
    model = model.to(DEV).eval()
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    infos, _ = synthetic.make_pool(1, seed=4, frames_per_scene=4)
    pool = PoolFrames.from_numpy([synthetic.make_point_cloud(60 + i, nsweeps=2) for i in range(4)], DEV)

    def loader():
        return DeviceSweepLoader(pool, cfg.voxel_generator, anchors, 2, device=DEV)

    feats, ent = sweep_embeddings(model, loader(), DEV, 4, with_entropy=True)
    # reference expression on the materialised detections
    ref_ent = []
    with torch.no_grad():
        for ex in loader():
            preds, _ = model(ex, return_loss=False, estimate=True)
            for p in preds:
                s = p["scores"]
                ref_ent.append((-s * torch.log(s) - (1.0 - s) * torch.log(1 - s)).mean())
    torch.testing.assert_close(ent, torch.stack(ref_ent), rtol=1e-5, atol=1e-6)
    ip, bp = str(tmp_path / "infos.pkl"), str(tmp_path / "buf.json")
    pickle.dump(infos, open(ip, "wb"))
    json.dump({"0": [], "1": [2]}, open(bp, "w"))
    common = dict(budget=1, buffer_file=bp, infos_origin=ip, pred=True, detector=model)
    badge = build_selector(dict(type="BadgeSelector", dataloader=loader(), weighted_feat_path="",
                                distance_store_file=None, **common))
    torch.testing.assert_close(badge.buffer_pred(local_rank=0), feats * ent[:, None], rtol=1e-6, atol=0)
    uwe = build_selector(dict(type="UWESelector", dataloader=loader(), weighted_feat_path="",
                              distance_store_file=None, **common))
    norm = (ent - ent.min()) / (ent.max() - ent.min())
    want = feats * norm[torch.tensor([0, 1, 0, 1], device=DEV)][:, None]      # batch-local index (quirk 8)
    torch.testing.assert_close(uwe.buffer_pred(local_rank=0), want, rtol=1e-5, atol=1e-7)
    es = build_selector(dict(type="EntropySelector", dataloader=loader(), buffer_path="", **common))
    es.select_samples(local_rank=0)
    assert es.selected_index["2"][-1] == 2 and len(es.selected_index["2"]) >= 2


def test_round2_sparse_entry_points_accept_empty_inputs():
    """Edge cases of the round-2 C entry points: zero rows (an empty level) must be a no-op, not a launch with a zero
    grid; bad io flags / channel counts are rejected with a message."""
    from al3d import detector_ops as D, lib
    from al3d.selector_ops import _ptr, _stream
    st = _stream()
    dummy = torch.zeros(8, device=DEV)
    idummy = torch.zeros(8, dtype=torch.int32, device=DEV)
    lib.call("al3d_sp_tile_ranges", _ptr(idummy), 0, 27, 0, _ptr(idummy), st)
    lib.call("al3d_sp_conv_rng_f16x3", _ptr(dummy), _ptr(idummy), 0, _ptr(idummy), _ptr(idummy), 27, _ptr(dummy), 32, 32,
             _ptr(dummy), None, None, 1, _ptr(dummy), 0, 0, st)
    lib.call("al3d_sp_conv_glds_f16x3_io", _ptr(dummy), _ptr(idummy), 0, _ptr(idummy), 27, _ptr(dummy), 32, 32, _ptr(dummy),
             None, None, 1, _ptr(dummy), 0, 3, st)
    assert D.rows_convert(torch.zeros((0, 32), device=DEV), True).shape == (0, 32)
    with pytest.raises(lib.Al3dError, match="multiple of 8"):
        D.rows_convert(torch.zeros((4, 12), device=DEV), True)
    with pytest.raises(lib.Al3dError, match="io flags"):
        lib.call("al3d_sp_conv_rng_f16x3", _ptr(dummy), _ptr(idummy), 256, _ptr(idummy), _ptr(idummy), 27, _ptr(dummy), 32, 32,
                 _ptr(dummy), None, None, 1, _ptr(dummy), 4, 9, st)
    with pytest.raises(lib.Al3dError, match="27-tap"):
        lib.call("al3d_sp_conv_rng_f16x3", _ptr(dummy), _ptr(idummy), 256, _ptr(idummy), _ptr(idummy), 3, _ptr(dummy), 32, 32,
                 _ptr(dummy), None, None, 1, _ptr(dummy), 4, 0, st)
    # pair rows survive a round trip through the module-level accessor
    from al3d.models.backbones import SparseTensor
    x = torch.randn(64, 32, device=DEV)
    sp = SparseTensor(D.rows_convert(x, True), torch.zeros((64, 4), dtype=torch.int32, device=DEV), [1, 8, 8], 1, pair_rows=True)
    assert float((sp.features - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    assert sp.features is sp.features                                  # converted once


@pytest.mark.parametrize("W", [2, 3, 11])                     # 2: narrower than a 12-byte probe (the per-(row, kz, ky) kernel runs)
@pytest.mark.parametrize("strided", [False, True])
def test_tiled_27_tap_table_equals_plain_table(W, strided):
    """`sp_table_rows27_kernel` (one thread per row, nine 12-byte probes, ballot-built tap masks) against the plain
    (row, kz, ky)-per-thread tables on a small, densely filled grid -- every border case of the clamped probes (x0 = -1,
    x0 = W - 2, rows beyond n) occurs -- and the tap masks against the table itself."""
    import ctypes
    from al3d import lib
    rng = np.random.default_rng(W * 2 + int(strided))
    B, D, H = 2, 5, 7
    cells = np.array([(b, z, y, x) for b in range(B) for z in range(D) for y in range(H) for x in range(W)], np.int32)
    coords = cells[rng.permutation(len(cells))[: max(3, int(0.7 * len(cells)))]].copy()
    n = len(coords)
    grid = torch.full((B * D * H * W,), -1, dtype=torch.int32, device=DEV)
    c_dev = _t(coords)
    st = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.call("al3d_sp_scatter_index", ptr(c_dev), n, B, D, H, W, ptr(grid), 1, st)
    i3 = lambda *v: (ctypes.c_int * 3)(*v)
    if strided:                                               # output sites: every cell of the stride-2 output grid
        OD, OH, OW = (D + 2 - 3) // 2 + 1, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        out = np.array([(b, z, y, x) for b in range(B) for z in range(OD) for y in range(OH) for x in range(OW)], np.int32)
        o_dev, m = _t(out), len(out)
        pitch = lib.load().al3d_sp_table_pitch(m)
        plain = torch.empty((27, m), dtype=torch.int32, device=DEV)
        tiled = torch.full((27, pitch), 12345, dtype=torch.int32, device=DEV)
        tmask = torch.full((pitch // 32,), -1, dtype=torch.int32, device=DEV)
        lib.call("al3d_sp_down_table", ptr(o_dev), m, i3(3, 3, 3), i3(2, 2, 2), i3(1, 1, 1), B, D, H, W, ptr(grid), ptr(plain), st)
        lib.call("al3d_sp_down_table_tiles", ptr(o_dev), m, i3(3, 3, 3), i3(2, 2, 2), i3(1, 1, 1), B, D, H, W, ptr(grid),
                 ptr(tiled), pitch, ptr(tmask), st)
    else:
        m = n
        pitch = lib.load().al3d_sp_table_pitch(m)
        plain = torch.empty((27, m), dtype=torch.int32, device=DEV)
        tiled = torch.full((27, pitch), 12345, dtype=torch.int32, device=DEV)
        tmask = torch.full((pitch // 32,), -1, dtype=torch.int32, device=DEV)
        lib.call("al3d_sp_subm_table", ptr(c_dev), m, B, D, H, W, ptr(grid), 3, 3, 3, ptr(plain), st)
        lib.call("al3d_sp_subm_table_tiles", ptr(c_dev), m, B, D, H, W, ptr(grid), 3, 3, 3, ptr(tiled), pitch, ptr(tmask), st)
    torch.cuda.synchronize()
    plain, tiled, tmask = plain.cpu().numpy(), tiled.cpu().numpy(), tmask.cpu().numpy().view(np.uint32)
    assert np.array_equal(tiled[:, :m], plain)
    assert np.all(tiled[:, m:] == -1)                         # rows beyond n: no neighbour
    assert (plain >= 0).any() and (plain < 0).any()
    want = np.zeros(pitch // 32, np.uint32)
    for k in range(27):
        want |= ((tiled[k].reshape(-1, 32) >= 0).any(1).astype(np.uint32) << np.uint32(k))
    assert np.array_equal(tmask, want)


@pytest.mark.parametrize("seed,score_bias,spread", [(21, 0.0, 0.25), (22, -1.0, 0.5), (23, 0.5, 1.0), (24, -2.0, 2.0), (25, 0.0, 4.0),
                                                    (26, 1.0, 0.1)])
def test_head_nms_super_rounds_equal_the_sequential_rule(oracle, seed, score_bias, spread):
    """The greedy loop runs in super-rounds of eight tentative survivors whose verdicts are resolved in rank order
    (csrc/head_nms.hip); the oracle applies the sequential rule one survivor at a time.  Box regressions of different
    spread around dense anchors give suppression patterns from "almost every tentative falls" (tight clusters) to
    "none does": kept boxes, their order, labels and scores must agree exactly / to rounding on every pattern."""
    B, H, W = 2, 16, 16
    cfg, head, anchors = _head_setup(seed, H, W, score_bias)
    head._prepare(torch.device(DEV))
    g = torch.Generator().manual_seed(seed)
    fused = torch.randn(B, H, W, head._ch, generator=g)
    for t in range(len(head.tasks)):                           # box regressions scaled: overlap structure changes with it
        b0 = head._box_off[t]
        fused[..., b0:b0 + head.num_anchor_per_locs[t] * 10] *= spread
        c0 = head._cls_off[t]
        fused[..., c0:c0 + head.num_anchor_per_locs[t] * head.num_classes[t]] += score_bias
    tc = cfg.test_cfg
    preds = [{"_fused": fused.to(DEV)}]
    out = head.predict({"anchors": [_t(a) for a in anchors], "metadata": [None] * B}, preds, tc)
    fz = fused.numpy().reshape(B, H * W, -1)
    label_off = np.concatenate([[0], np.cumsum(head.num_classes)])
    total = 0
    for b in range(B):
        bb, ss, ll = [], [], []
        for t in range(len(head.tasks)):
            bx, sc, lb = oracle.head_predict(fz[b], anchors[t], head.num_anchor_per_locs[t], head.num_classes[t],
                                             head._box_off[t], head._cls_off[t], tc.score_threshold,
                                             tc.nms.nms_iou_threshold, tc.nms.nms_pre_max_size, tc.nms.nms_post_max_size,
                                             tc.post_center_limit_range)
            bb.append(bx); ss.append(sc); ll.append(lb + label_off[t])
        bb, ss, ll = np.concatenate(bb), np.concatenate(ss), np.concatenate(ll)
        got = out[b]
        assert got["label_preds"].cpu().numpy().tolist() == ll.tolist()
        np.testing.assert_allclose(got["scores"].cpu().numpy(), ss, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got["box3d_lidar"].cpu().numpy()[:, :8], bb[:, :8], rtol=1e-4, atol=1e-4)
        total += len(ss)
    assert total > 0
