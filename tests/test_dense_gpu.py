"""GPU suite: MFMA dense conv / deconv / GAP kernels against a plain PyTorch fp32 CPU
reference of the same op (floating point => tolerance, stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32 products, fp32 accumulation in a different order than the CPU reference:
# |err| <= ~K * eps * sum|a*b|; K <= 2304 here.
RTOL, ATOL = 2e-5, 2e-5


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,relu", [
    (1, 16, 16, 32, 128, 3, 1, 1, True),
    (2, 24, 40, 64, 128, 3, 1, 1, True),      # ragged patches (24x40 not multiples of 8x16)
    (1, 32, 32, 128, 256, 3, 2, 1, True),     # block1 entry: ZeroPad2d(1)+conv stride 2
    (2, 13, 19, 32, 40, 1, 1, 0, False),      # head-style 1x1, Cout not a multiple of 128
    (1, 128, 128, 256, 128, 3, 1, 1, True),   # block0 entry at full BEV size
])
def test_conv2d_matches_torch(B, H, W, Cin, Cout, k, s, p, relu):
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(H * 1000 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w, stride=s, padding=p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        ref = ref.relu()
    got = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_conv_weight(w).to(DEV), scale.to(DEV), shift.to(DEV),
                        k, s, p, relu)
    torch.testing.assert_close(got.cpu(), _nhwc(ref), rtol=RTOL, atol=ATOL)


def test_conv2d_identity_asymmetric():
    """A = I check with an asymmetric B: catches transposed C/D register maps."""
    from al3d import detector_ops as D
    Cin = Cout = 64
    x = torch.zeros(1, Cin, 8, 16)
    for c in range(Cin):
        x[0, c, c % 8, (3 * c) % 16] = 1.0 + c
    w = torch.arange(Cout * Cin, dtype=torch.float32).view(Cout, Cin, 1, 1) / 100.0
    ref = F.conv2d(x, w)
    got = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_conv_weight(w).to(DEV), None, None, 1, 1, 0, False)
    torch.testing.assert_close(got.cpu(), _nhwc(ref), rtol=1e-6, atol=1e-6)


def test_deconv_and_concat_window():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 12, 20, generator=g)
    w = torch.randn(64, 96, 2, 2, generator=g) / 8.0
    scale = torch.rand(96, generator=g) + 0.5
    shift = torch.randn(96, generator=g) * 0.1
    ref = (F.conv_transpose2d(x, w, stride=2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).relu()
    out = torch.full((2, 24, 40, 160), -7.0, device=DEV)
    D.deconv2x2_nhwc(_nhwc(x).to(DEV), D.pack_deconv_weight(w).to(DEV), scale.to(DEV), shift.to(DEV),
                     True, out=out, coff=64)
    o = out.cpu()
    torch.testing.assert_close(o[..., 64:], _nhwc(ref), rtol=RTOL, atol=ATOL)
    assert torch.all(o[..., :64] == -7.0)          # channel window respected


def test_gap_matches_two_stage_mean():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 100, 24, 40, generator=g)     # NCHW
    ref = x.mean(-1).mean(-1)
    got = D.gap_nhwc(_nhwc(x).to(DEV))
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- bf16x6 (fp32-faithful) kernels
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(1, 16, 16, 32, 128, 3, 1, 1), (2, 24, 40, 64, 128, 3, 1, 1),
                                                  (1, 32, 32, 128, 256, 3, 2, 1), (2, 13, 19, 32, 40, 1, 1, 0),
                                                  (1, 64, 64, 256, 128, 3, 1, 1)])
def test_conv2d_bf16x6_is_fp32_faithful(B, H, W, Cin, Cout, k, s, p):
    """The six-product bf16 split must be as close to the exact (fp64) convolution as the
    fp32-input MFMA kernel is: compare both against an fp64 CPU reference."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(H * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g))  # wide dynamic range
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p)
    ref = _nhwc(ref)
    wp = D.pack_conv_weight(w).to(DEV)
    got32 = D.conv2d_nhwc(_nhwc(x).to(DEV), wp, None, None, k, s, p, False).cpu().double()
    got6 = D.conv2d_nhwc(_nhwc(x).to(DEV), D.split_bf16x3(wp), None, None, k, s, p, False).cpu().double()
    scale = F.conv2d(x.abs().double(), w.abs().double(), stride=s, padding=p)   # sum |a*b| per output
    scale = _nhwc(scale)
    e32 = ((got32 - ref).abs() / scale).max().item()
    e6 = ((got6 - ref).abs() / scale).max().item()
    # both are fp32-accumulation noise (K up to 2304): below 1.5e-6 of sum|a*b|; the split may not
    # be worse than 2x the fp32-input kernel (measured: it is slightly better)
    assert e32 < 1.5e-6 and e6 < 1.5e-6 and e6 < 2.0 * e32 + 1e-8, (e32, e6)


def test_split_bf16x3_is_exact():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(4096, generator=g) * torch.exp(torch.randn(4096, generator=g) * 3)).to(DEV)
    s3 = D.split_bf16x3(w).float()
    assert torch.equal(s3[0] + s3[1] + s3[2], w)          # x = x1 + x2 + x3 exactly
    assert torch.equal(s3[0], w.bfloat16().float())       # hi piece = RNE bf16


def test_deconv_bf16x6_matches_f32_kernel():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 12, 20, generator=g)
    w = torch.randn(64, 96, 2, 2, generator=g) / 8.0
    wp = D.pack_deconv_weight(w).to(DEV)
    a = D.deconv2x2_nhwc(_nhwc(x).to(DEV), wp, None, None, False)
    b = D.deconv2x2_nhwc(_nhwc(x).to(DEV), D.split_bf16x3(wp), None, None, False)
    torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6)


# ---------------------------------------------------------------- f16x3 (fp32-class) kernels
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(1, 16, 16, 32, 128, 3, 1, 1), (2, 24, 40, 64, 128, 3, 1, 1),
                                                  (1, 32, 32, 128, 256, 3, 2, 1), (2, 13, 19, 32, 40, 1, 1, 0),
                                                  (1, 64, 64, 256, 128, 3, 1, 1)])
@pytest.mark.parametrize("xmag", [1.0, 1e-4, 300.0])
def test_conv2d_f16x3_is_fp32_class(B, H, W, Cin, Cout, k, s, p, xmag):
    """Three f16 products per MAC: error against the exact (fp64) convolution, as a fraction of
    sum|a*b|, must stay at the fp32-input kernel's level -- at O(1) activations, at activations
    below f16's normal range (1e-4 * lognormal: the high piece of many values is an f16 subnormal, the lifted residual piece
    carries the rest) and at large ones (300 * lognormal, still < 65504).  Tolerance: 1.5e-6 of sum|a*b| like the bf16x6
    test, and at most 3x the fp32-input kernel's error (measured ~1x)."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(H * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g)) * xmag
    x = x.clamp(-6.0e4, 6.0e4)                       # the lognormal tail may not leave f16's range
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = _nhwc(F.conv2d(x.double(), w.double(), stride=s, padding=p))
    wp = D.pack_conv_weight(w).to(DEV)
    got32 = D.conv2d_nhwc(_nhwc(x).to(DEV), wp, None, None, k, s, p, False).cpu().double()
    w3, sc3 = D.split_f16x3(wp)
    got3 = D.conv2d_nhwc(_nhwc(x).to(DEV), w3, sc3, None, k, s, p, False)
    if D.frag_ok(Cout, Cin, k, s, p):
        # same arithmetic with the weights streamed in fragment order: the same bits
        gotf = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_frag_f16x3(w3), sc3, None, k, s, p, False)
        assert torch.equal(gotf, got3)
    if (k, s, p) != (3, 1, 1) or Cin % 32:            # the geometries the generic kernels serve
        gotb = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_bstream_f16x3(w3), sc3, None, k, s, p, False)
        assert torch.equal(gotb, got3)                # streamed weights: same products, same order
    got16 = None
    if D.frag_ok(Cout, Cin, k, s, p) and Cin % 64 == 0:
        # the 16x16x32 kernel: the same three products, 32 input channels per instruction -> another summation
        # grouping (not the same bits), the same error class
        got16 = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_frag16_f16x3(w3), sc3, None, k, s, p, False)
        again = D.conv2d_nhwc(_nhwc(x).to(DEV), D.pack_frag16_f16x3(w3), sc3, None, k, s, p, False)
        assert torch.equal(got16, again)              # deterministic
        got16 = got16.cpu().double()
    got3 = got3.cpu().double()
    scale = _nhwc(F.conv2d(x.abs().double(), w.abs().double(), stride=s, padding=p))
    e32 = ((got32 - ref).abs() / scale).max().item()
    e3 = ((got3 - ref).abs() / scale).max().item()
    assert e32 < 1.5e-6 and e3 < 1.5e-6 and e3 < 3.0 * e32 + 1e-8, (e32, e3)
    if got16 is not None:
        e16 = ((got16 - ref).abs() / scale).max().item()
        assert e16 < 1.5e-6 and e16 < 3.0 * e32 + 1e-8, (e32, e16)


def test_split_f16x3_planes():
    """wh + wl reproduces w * 2^s to 2^-22 relative, max|w 2^s| in (2^13, 2^14]."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 64, generator=g) * 0.03).to(DEV)
    pl, sc = D.split_f16x3(w)
    mul = 1.0 / float(sc[0])
    assert mul == 2.0 ** round(np.log2(mul))
    ws = w.double() * mul
    assert 2.0 ** 13 < float(ws.abs().max()) <= 2.0 ** 14
    p = pl.double()
    big = ws.abs() > 0.25                                   # lo piece in f16's normal range
    assert float(((p[0] + p[1] - ws).abs() / ws.abs())[big].max()) < 2.0 ** -22
    assert float((p[0] + p[1] - ws).abs().max()) < 2.0 ** -9    # absolute, in units of 2^-s
    assert pl.shape == (2, 64, 64)


def test_conv2d_f16x3_scale_shift_relu_and_window():
    """BN scale (with the folded weight exponent), shift, ReLU and the concat window."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 24, 40, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    scale = torch.rand(96, generator=g) + 0.5
    shift = torch.randn(96, generator=g) * 0.1
    ref = (F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).relu()
    w3, sc3 = D.split_f16x3(D.pack_conv_weight(w).to(DEV), scale.to(DEV))
    out = torch.full((2, 24, 40, 160), -7.0, device=DEV)
    D.conv2d_nhwc(_nhwc(x).to(DEV), w3, sc3, shift.to(DEV), 3, 1, 1, True, out=out, coff=64)
    o = out.cpu()
    torch.testing.assert_close(o[..., 64:], _nhwc(ref), rtol=RTOL, atol=ATOL)
    assert torch.all(o[..., :64] == -7.0)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 128, 128, 256, 128), (2, 24, 40, 64, 128), (3, 9, 70, 128, 256),
                                            (1, 64, 64, 256, 256)])
def test_conv3x3_f16x3_fragment_stream_equals_lds_staged_kernel(B, H, W, Cin, Cout):
    """Both 3x3 kernels run the same products in the same order: bit-identical outputs, including
    ragged tiles (H, W not multiples of 4 x 32), scale/shift/ReLU and a concat window."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 9, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    a = torch.full((B, H, W, Cout + 32), -3.0, device=DEV)
    b = torch.full((B, H, W, Cout + 32), -3.0, device=DEV)
    D.conv2d_nhwc(x, w3, sc3, shift, 3, 1, 1, True, out=a, coff=32)
    D.conv2d_nhwc(x, D.pack_frag_f16x3(w3), sc3, shift, 3, 1, 1, True, out=b, coff=32)
    assert torch.equal(a, b)
    assert torch.all(b[..., :32] == -3.0)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(2, 128, 128, 128, 256, 1, 1, 0),    # deblock 0
                                                  (1, 128, 128, 128, 256, 3, 2, 1),    # block 1 entry
                                                  (2, 37, 21, 512, 236, 1, 1, 0),      # fused head, ragged
                                                  (1, 19, 50, 64, 40, 3, 1, 0),        # 3x3 without padding
                                                  (1, 9, 11, 48, 16, 1, 1, 0),         # 3 steps: not a multiple of the 6-step unroll
                                                  (3, 10, 9, 32, 300, 5, 2, 2)])       # 25 taps, 3 column blocks
def test_conv2d_f16x3_streamed_weights_equal_lds_staged_kernel(B, H, W, Cin, Cout, k, s, p):
    """The streamed-weight generic kernel walks the same (tap, chunk) steps with the same three
    products per step: outputs are bit-identical to the LDS-staged generic kernel (incl. Cout
    padding, ragged tiles, padding taps, scale/shift/ReLU and the concat window)."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(Cin + H + k)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, k * k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    a = torch.full((B, OH, OW, Cout + 20), -3.0, device=DEV)
    b = torch.full((B, OH, OW, Cout + 20), -3.0, device=DEV)
    D.conv2d_nhwc(x, w3, sc3, shift, k, s, p, True, out=a, coff=20)
    D.conv2d_nhwc(x, D.pack_bstream_f16x3(w3), sc3, shift, k, s, p, True, out=b, coff=20)
    assert torch.equal(a, b)
    assert torch.all(b[..., :20] == -3.0)
    # both operands by LDS-DMA, split on the fragment: the same bits again, and the same fused-GAP partials
    c = torch.full((B, OH, OW, Cout + 20), -3.0, device=DEV)
    parts = D.gap_parts(OH, OW, False)
    ga = torch.full((B, parts + 3, Cout + 20), -7.0, device=DEV)
    gc = torch.full((B, parts + 3, Cout + 20), -7.0, device=DEV)
    D.conv2d_nhwc(x, w3, sc3, shift, k, s, p, True, out=a, coff=20, gap=ga)
    D.conv2d_nhwc(x, D.pack_dma_f16x3(w3), sc3, shift, k, s, p, True, out=c, coff=20, gap=gc)
    assert torch.equal(a, c) and torch.equal(ga, gc)
    c.fill_(-3.0)
    D.conv2d_nhwc(x, D.pack_dma_f16x3(w3), sc3, shift, k, s, p, True, out=c, coff=20)      # without the GAP
    assert torch.equal(a, c)


def test_deconv_f16x3_streamed_weights_equal_lds_staged_kernel():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 33, 20, 256, generator=g).to(DEV)
    w = (torch.randn(200, 4, 256, generator=g) / 32.0).to(DEV)
    scale = (torch.rand(200, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(200, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    a = torch.full((2, 66, 40, 264), -3.0, device=DEV)
    b = torch.full((2, 66, 40, 264), -3.0, device=DEV)
    D.deconv2x2_nhwc(x, w3, sc3, shift, True, out=a, coff=64)
    D.deconv2x2_nhwc(x, D.pack_bstream_f16x3(w3), sc3, shift, True, out=b, coff=64)
    assert torch.equal(a, b)
    assert torch.all(b[..., :64] == -3.0)
    c = torch.full((2, 66, 40, 264), -3.0, device=DEV)
    parts = D.gap_parts(66, 40, True)
    ga = torch.full((2, parts, 264), -7.0, device=DEV)
    gc = torch.full((2, parts, 264), -7.0, device=DEV)
    D.deconv2x2_nhwc(x, w3, sc3, shift, True, out=a, coff=64, gap=ga)
    D.deconv2x2_nhwc(x, D.pack_dma_f16x3(w3), sc3, shift, True, out=c, coff=64, gap=gc)
    assert torch.equal(a, c) and torch.equal(ga, gc)


def test_deconv_f16x3_matches_f32_kernel():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 12, 20, generator=g)
    w = torch.randn(64, 96, 2, 2, generator=g) / 8.0
    wp = D.pack_deconv_weight(w).to(DEV)
    a = D.deconv2x2_nhwc(_nhwc(x).to(DEV), wp, None, None, False)
    w3, sc3 = D.split_f16x3(wp)
    b = D.deconv2x2_nhwc(_nhwc(x).to(DEV), w3, sc3, None, False)
    torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6)


def test_conv2d_f16x3_out_of_range_activation_is_not_silent():
    """|x| >= 65504 cannot be carried by the f16 pieces: the output must be non-finite, never a
    plausible wrong number (the sweep checks its embeddings for finiteness)."""
    from al3d import detector_ops as D
    x = torch.ones(1, 8, 8, 32)
    x[0, 3, 3, 5] = 7.0e4
    w3, sc3 = D.split_f16x3(torch.full((128, 9, 32), 0.01).to(DEV))
    out = D.conv2d_nhwc(x.to(DEV), w3, sc3, None, 3, 1, 1, False).cpu()
    assert not torch.isfinite(out[0, 3, 3]).any()
    assert torch.isfinite(out[0, 7, 7]).all()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(2, 24, 40, 128, 256, 1, 1, 0), (1, 33, 20, 128, 256, 3, 2, 1),
                                                  (2, 37, 21, 512, 236, 1, 1, 0), (1, 19, 50, 64, 40, 3, 1, 0)])
def test_conv2d_dma_pair_pixels(B, H, W, Cin, Cout, k, s, p):
    """Pair pixels on the LDS-DMA dense kernel (csrc/sp_rows.h format on NHWC maps): (1) a layer fed pair pixels
    gives the bits of the layer fed the f32 map they were split from; (2) writing pair pixels = splitting the f32
    output (untouched concat-window channels stay untouched); (3) the fused-GAP partials do not depend on the output
    format; (4) both together compose."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(Cin + H + k)
    x = (torch.randn(B, H, W, Cin, generator=g) * torch.exp(torch.randn(B, H, W, Cin, generator=g))).to(DEV)
    w = (torch.randn(Cout, k * k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    wd = D.pack_dma_f16x3(w3)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    xp = D.rows_convert(x.view(-1, Cin), True).view_as(x)
    parts = D.gap_parts(OH, OW, False)
    ldc = Cout + 24 if Cout % 8 == 0 else Cout

    def run(xin, io, gap):
        out = torch.full((B, OH, OW, ldc), -3.0, device=DEV)
        D.conv2d_nhwc(xin, wd, sc3, shift, k, s, p, True, out=out, coff=ldc - Cout, gap=gap, io=io)
        return out
    g0 = torch.zeros(B, parts, ldc, device=DEV)
    base = run(x, 0, g0)
    g1 = torch.zeros_like(g0)
    assert torch.equal(run(xp, D.IO_IN_PAIR, g1), base) and torch.equal(g1, g0)                  # (1)
    if Cout % 8 == 0:
        g2 = torch.zeros_like(g0)
        outp = run(x, D.IO_OUT_PAIR, g2)
        assert torch.equal(g2, g0)                                                               # (3)
        assert torch.all(outp[..., :24] == -3.0)
        want = D.rows_convert(base[..., 24:].reshape(-1, Cout).contiguous(), True)
        assert torch.equal(outp[..., 24:].reshape(-1, Cout).view(torch.int32), want.view(torch.int32))      # (2)
        both = run(xp, D.IO_IN_PAIR | D.IO_OUT_PAIR, torch.zeros_like(g0))
        assert torch.equal(both.view(torch.int32), outp.view(torch.int32))                       # (4)
    else:
        with pytest.raises(Exception, match="multiples of 8"):
            run(x, D.IO_OUT_PAIR, None)


def test_deconv_dma_pair_pixels():
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 33, 20, 256, generator=g).to(DEV)
    w = (torch.randn(256, 4, 256, generator=g) / 32.0).to(DEV)
    scale = (torch.rand(256, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(256, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    wd = D.pack_dma_f16x3(w3)
    xp = D.rows_convert(x.view(-1, 256), True).view_as(x)
    parts = D.gap_parts(66, 40, True)

    def run(xin, io):
        out = torch.full((2, 66, 40, 512), -3.0, device=DEV)
        gap = torch.zeros(2, parts, 512, device=DEV)
        D.deconv2x2_nhwc(xin, wd, sc3, shift, True, out=out, coff=256, gap=gap, io=io)
        return out, gap
    base, g0 = run(x, 0)
    a, g1 = run(xp, D.IO_IN_PAIR)
    assert torch.equal(a, base) and torch.equal(g1, g0)
    b, g2 = run(xp, D.IO_IN_PAIR | D.IO_OUT_PAIR)
    assert torch.equal(g2, g0) and torch.all(b[..., :256] == -3.0)
    want = D.rows_convert(base[..., 256:].reshape(-1, 256).contiguous(), True)
    assert torch.equal(b[..., 256:].reshape(-1, 256).view(torch.int32), want.view(torch.int32))


def test_conv3x3_streamed_kernel_writes_pair_pixels():
    """The streamed 3x3 kernel's pair-pixel epilogue (io = 2): the stored map is the split of the f32 output, the
    channels outside the concat window stay untouched, ragged tiles included."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 37, 45, 64, generator=g).to(DEV)
    w = (torch.randn(128, 9, 64, generator=g) / 24.0).to(DEV)
    scale = (torch.rand(128, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(128, generator=g) * 0.1).to(DEV)
    w3, sc3 = D.split_f16x3(w, scale)
    wf = D.pack_frag_f16x3(w3)
    a = torch.full((2, 37, 45, 160), -3.0, device=DEV)
    b = torch.full((2, 37, 45, 160), -3.0, device=DEV)
    D.conv2d_nhwc(x, wf, sc3, shift, 3, 1, 1, True, out=a, coff=32)
    D.conv2d_nhwc(x, wf, sc3, shift, 3, 1, 1, True, out=b, coff=32, io=D.IO_OUT_PAIR)
    assert torch.all(b[..., :32] == -3.0)
    want = D.rows_convert(a[..., 32:].reshape(-1, 128).contiguous(), True)
    assert torch.equal(b[..., 32:].reshape(-1, 128).view(torch.int32), want.view(torch.int32))
    with pytest.raises(Exception, match="reads f32"):
        D.conv2d_nhwc(x, wf, sc3, shift, 3, 1, 1, True, out=b, coff=32, io=D.IO_IN_PAIR)


@pytest.mark.parametrize("B,H,W,Cin,Cout,xmag", [(2, 32, 32, 128, 128, 1.0), (1, 37, 21, 256, 128, 1e-4), (2, 16, 48, 256, 256, 300.0),
                                                 (1, 64, 64, 128, 64, 1.0)])
def test_conv3x3_winograd_f16x3_is_fp32_class(B, H, W, Cin, Cout, xmag):
    """Winograd F(2x2, 3x3) in f16x3 arithmetic against the exact (fp64) convolution: the same gate as
    test_conv2d_f16x3_is_fp32_class -- <= 1.5e-6 of sum|a*b| and at most 3x the fp32-input kernel's error -- at O(1),
    tiny and large activations, on maps that are not multiples of the 16 x 16 tile; pair-pixel output holds the same values;
    BN scale / shift / ReLU and the channel window."""
    from al3d import detector_ops as D
    g = torch.Generator().manual_seed(H * 7 + Cin)
    x = (torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g)) * xmag).clamp(-6.0e4, 6.0e4)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    ref = _nhwc(F.conv2d(x.double(), w.double(), padding=1))
    scale = _nhwc(F.conv2d(x.abs().double(), w.abs().double(), padding=1))
    wp = D.pack_conv_weight(w).to(DEV)
    xd = _nhwc(x).to(DEV)
    got32 = D.conv2d_nhwc(xd, wp, None, None, 3, 1, 1, False).cpu().double()
    ww, sc = D.pack_wino_f16x3(wp)
    got = D.conv2d_nhwc(xd, ww, sc, None, 3, 1, 1, False)
    assert torch.equal(got, D.conv2d_nhwc(xd, ww, sc, None, 3, 1, 1, False))          # deterministic
    e32 = ((got32 - ref).abs() / scale).max().item()
    ew = ((got.cpu().double() - ref).abs() / scale).max().item()
    assert ew < 1.5e-6 and ew < 3.0 * e32 + 1e-8, (e32, ew)
    pair = D.conv2d_nhwc(xd, ww, sc, None, 3, 1, 1, False, io=D.IO_OUT_PAIR)
    back = D.rows_convert(pair.view(-1, Cout), False).view_as(got)
    assert bool(((back - got).abs() <= 2.0 ** -21 * got.abs() + 2.0 ** -35).all())
    # epilogue: BN scale / shift / ReLU into a channel window of a wider map
    bs, bt = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    ww2, sc2 = D.pack_wino_f16x3(wp, bs)
    wide = torch.full((B, H, W, Cout + 16), 7.0, device=DEV)
    D.conv2d_nhwc(xd, ww2, sc2, bt, 3, 1, 1, True, out=wide, coff=8)
    want = torch.relu(ref.to(DEV) * bs.double() + bt.double())
    assert float((wide[..., 8:8 + Cout].double() - want).abs().max()) <= 2e-6 * float(scale.max()) * float(bs.max()) + 1e-6
    assert bool((wide[..., :8] == 7.0).all()) and bool((wide[..., 8 + Cout:] == 7.0).all())
