"""Per-layer time of the view transform's convolutions at B samples (HIP events, 5 reps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from al3d import synthetic
from al3d.models.bevfusion_camera import DepthLSSTransform
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
vt = DepthLSSTransform(256, 80, (256, 704), (32, 88), [-54.0, 54.0, 0.3], [-54.0, 54.0, 0.3], [-10.0, 10.0, 20.0], [1.0, 60.0, 0.5], downsample=2)
synthetic.seed_modules_(vt, 3)
vt = vt.to(dev).eval()
def timeit(f, x, n=5):
    y = f(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = f(x)
    e1.record(); torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) / n
with torch.no_grad():
    d = torch.rand(B * 6, 256, 704, 1, device=dev)
    for i, l in enumerate(vt._dt):
        d2, t = timeit(l, d); print("dtransform", i, tuple(d.shape), "->", tuple(d2.shape), round(t, 3), "ms"); d = d2
    y = torch.randn(B * 6, 32, 88, 320, device=dev)
    for i, l in enumerate(vt._dn):
        y2, t = timeit(l, y); print("depthnet", i, tuple(y.shape), "->", tuple(y2.shape), round(t, 3), "ms"); y = y2
    x = torch.randn(B, 360, 360, 80, device=dev)
    for i, l in enumerate(vt._ds):
        x2, t = timeit(l, x); print("downsample", i, tuple(x.shape), "->", tuple(x2.shape), round(t, 3), "ms"); x = x2
