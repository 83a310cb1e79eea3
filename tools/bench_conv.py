"""Micro-benchmark of the MFMA conv kernel on the RPN layer shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from al3d import detector_ops as D

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else 'f32'
shapes = [  # H, Cin, Cout, k, s, p
    (128, 256, 128, 3, 1, 1), (128, 128, 128, 3, 1, 1), (128, 128, 256, 1, 1, 0),
    (128, 128, 256, 3, 2, 1), (64, 256, 256, 3, 1, 1), (128, 512, 236, 1, 1, 0)]
for (H, Cin, Cout, k, s, p) in shapes:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, k * k, Cin, device=dev) * 0.05
    if os.environ.get("AL3D_BENCH_ZERO") == "1":      # data-dependent power: all-zero operands toggle nothing
        x.zero_(); w.zero_()
    if MODE == "bf16x6":
        w = D.split_bf16x3(w)
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    if MODE == 'f16x3':
        w, sc = D.split_f16x3(w, sc)
    if MODE == 'f16x3frag16':
        w, sc = D.split_f16x3(w, sc)
        w = D.pack_frag16_f16x3(w) if (D.frag_ok(Cout, Cin, k, s, p) and Cin % 64 == 0) else D.pack_bstream_f16x3(w)
    io = 0
    if MODE == 'f16x3fragpair' and D.frag_ok(Cout, Cin, k, s, p):
        io = D.IO_OUT_PAIR
    if MODE == 'f16x3dmapair' and not D.frag_ok(Cout, Cin, k, s, p):
        x = D.rows_convert(x.view(-1, Cin), True).view_as(x)
        io = D.IO_IN_PAIR | (D.IO_OUT_PAIR if Cout % 8 == 0 else 0)
    if MODE in ('f16x3dma', 'f16x3dmapair', 'f16x3fragpair'):
        w, sc = D.split_f16x3(w, sc)
        w = D.pack_frag_f16x3(w) if D.frag_ok(Cout, Cin, k, s, p) else D.pack_dma_f16x3(w)
    if MODE == 'f16x3wino':
        if D.wino_ok(Cout, Cin, k, s, p):
            w, sc = D.pack_wino_f16x3(w, sc)
        else:
            w, sc = D.split_f16x3(w, sc)
            w = D.pack_dma_f16x3(w)
    if MODE == 'f16x3frag':
        w, sc = D.split_f16x3(w, sc)
        w = D.pack_frag_f16x3(w) if D.frag_ok(Cout, Cin, k, s, p) else D.pack_bstream_f16x3(w)
    OH = (H + 2 * p - k) // s + 1
    out = torch.empty(B, OH, OH, Cout, device=dev)
    for _ in range(3):
        D.conv2d_nhwc(x, w, sc, sh, k, s, p, True, out=out, io=io)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        D.conv2d_nhwc(x, w, sc, sh, k, s, p, True, out=out, io=io)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * OH * OH * Cout * Cin * k * k
    print(f"{MODE} B={B} H={H} Cin={Cin} Cout={Cout} k={k} s={s}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")

# the 2x2 transposed convolution of the second deblock (rpn.py:124-142): [B,64,64,256] -> [B,128,128,256]
if MODE.startswith("f16x3"):
    x = torch.randn(B, 64, 64, 256, device=dev)
    w = torch.randn(256, 4, 256, device=dev) * 0.05
    sc = torch.ones(256, device=dev); sh = torch.zeros(256, device=dev)
    w, sc = D.split_f16x3(w, sc)
    dio = 0
    if MODE in ("f16x3dma", "f16x3dmapair"):
        w = D.pack_dma_f16x3(w)
        if MODE == "f16x3dmapair":
            x = D.rows_convert(x.view(-1, 256), True).view_as(x)
            dio = D.IO_IN_PAIR | D.IO_OUT_PAIR
    elif MODE != "f16x3":
        w = D.pack_bstream_f16x3(w)
    out = torch.empty(B, 128, 128, 512, device=dev)
    for _ in range(3):
        D.deconv2x2_nhwc(x, w, sc, sh, True, out=out, coff=256, io=dio)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        D.deconv2x2_nhwc(x, w, sc, sh, True, out=out, coff=256, io=dio)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * B * 64 * 64 * 4 * 256 * 256
    print(f"{MODE} B={B} deconv 2x2 256->256: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
