"""Dev tool: per-stage time of the camera-branch slice at BASELINE configs[4] shapes (one sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from al3d import detector_ops as D, synthetic
from al3d.models.bevfusion_camera import ConvFuser, LSSViewTransform, bev_pool
from al3d.models.necks import RPN
from test_bevpool_gpu import _calib
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, C = 6, 80
vt = LSSViewTransform(C, (256, 704), (32, 88), (-54.0, 54.0, 0.3), (-54.0, 54.0, 0.3), (-10.0, 10.0, 20.0), (1.0, 60.0, 0.5), 2)
fuser = ConvFuser([80, 256], 256)
dec = RPN(layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256], us_layer_strides=[1, 2],
          us_num_filters=[256, 256], num_input_features=256)
for i, m in enumerate((vt.downsample, fuser, dec)):
    synthetic.seeded_init_(m, seed=10 + i)
vt, fuser, dec = vt.to(DEV).eval(), fuser.to(DEV).eval(), dec.to(DEV).eval()
rots, trans, intr, prot, ptr_ = _calib(B, N, np.random.default_rng(8))
depth = torch.softmax(torch.randn(B, N, 118, 32, 88, device=DEV), dim=2)
ctx = torch.randn(B, N, 32, 88, C, device=DEV) * 0.5
lidar = torch.relu(torch.randn(B, 180, 180, 256, device=DEV))
def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms"); return r
with torch.no_grad():
    timed("geometry (torch, reference)", lambda: vt.get_geometry(rots, trans, intr, prot, ptr_).contiguous(), reps=2)
    geom = timed("geometry (device kernel)", lambda: vt.geometry_device(rots, trans, intr, prot, ptr_))
    pooled = timed("bev_pool (fused LSS)", lambda: bev_pool(ctx.reshape(B * N, 32, 88, C).contiguous(), geom, B, vt.dx.cpu().numpy(),
                   vt.bx.cpu().numpy(), vt.nx.cpu().numpy(), depth=depth.reshape(B * N, 118, 32, 88).contiguous()))
    def ds(x):
        for l in vt._ds: x = l(x)
        return x
    cam = timed("downsample convs (3)", lambda: ds(pooled))
    fused = timed("ConvFuser", lambda: fuser([cam, lidar]))
    neck = timed("decoder (SECOND+FPN)", lambda: dec(fused))
    timed("GAP", lambda: D.gap_nhwc(neck))
