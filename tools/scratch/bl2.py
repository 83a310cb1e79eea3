import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
for b in (16, 32, 64):
    r = bench.bevfusion_lidar_leg(dev, frames=3 * b, batch=b)
    print(b, r["value"], flush=True)
