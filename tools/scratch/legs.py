import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
r = bench.bevfusion_lidar_leg(dev); print("lidar", r["value"], flush=True)
r = bench.bevfusion_camera_lidar_leg(dev); print("camera+lidar", r["value"], flush=True)
