import sys, torch, json
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
for b in (12, 16, 24):
    r = bench.bevfusion_camera_lidar_leg(dev, frames=3 * b, batch=b)
    print(b, r["value"], r["ms_per_sample"], flush=True)
