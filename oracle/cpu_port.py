"""TEST INFRASTRUCTURE: the CPU leg of bench.py (`cpu_baseline`, kind "port").

Runs the oracle's plain-C restatements (voxelize, sparse encoder, rotated NMS, selector maps,
greedy) plus torch-CPU fp32 convs for the dense neck/head on the host cores, on a bounded
sample, and converts to the bench metric.  Never imported by the product package.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

import oracle


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def _fold(bn):
    inv = 1.0 / np.sqrt(bn["running_var"].numpy().astype(np.float64) + 1e-3)
    scale = bn["weight"].numpy() * inv
    shift = bn["bias"].numpy() - bn["running_mean"].numpy() * scale
    return scale.astype(np.float32), shift.astype(np.float32)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def sparse_encoder(sd, feats, coords, shape):
    """FPNSpMiddleResNetFHD with the oracle's spconv (one frame)."""
    stages = [("backbone.middle_conv0.", [("c", 0, 1, None), ("b", 3), ("b", 4), ("d", 5, 6, (3, 3, 3), (2, 2, 2), (1, 1, 1))]),
              ("backbone.middle_conv1.", [("b", 0), ("b", 1), ("d", 2, 3, (3, 3, 3), (2, 2, 2), (1, 1, 1))]),
              ("backbone.middle_conv2.", [("b", 0), ("b", 1), ("d", 2, 3, (3, 3, 3), (2, 2, 2), (0, 1, 1))]),
              ("backbone.middle_conv3.", [("b", 0), ("b", 1), ("d", 2, 3, (3, 1, 1), (2, 1, 1), (0, 0, 0))])]
    for prefix, items in stages:
        for it in items:
            if it[0] == "c":
                w = sd[f"{prefix}{it[1]}.weight"].numpy()
                sc, sh = _fold(_sub(sd, f"{prefix}{it[2]}."))
                f, coords, shape = oracle.spconv(feats, coords, 1, shape, w, (3, 3, 3), (1, 1, 1), (0, 0, 0), True)
                feats = np.maximum(f * sc + sh, 0)
            elif it[0] == "b":
                p = f"{prefix}{it[1]}."
                idt = feats
                sc, sh = _fold(_sub(sd, p + "bn1."))
                f, _, _ = oracle.spconv(feats, coords, 1, shape, sd[p + "conv1.weight"].numpy(), (3, 3, 3), (1, 1, 1), (0, 0, 0), True)
                f = np.maximum((f + sd[p + "conv1.bias"].numpy()) * sc + sh, 0)
                sc, sh = _fold(_sub(sd, p + "bn2."))
                f, _, _ = oracle.spconv(f, coords, 1, shape, sd[p + "conv2.weight"].numpy(), (3, 3, 3), (1, 1, 1), (0, 0, 0), True)
                feats = np.maximum((f + sd[p + "conv2.bias"].numpy()) * sc + sh + idt, 0)
            else:
                w = sd[f"{prefix}{it[1]}.weight"].numpy()
                sc, sh = _fold(_sub(sd, f"{prefix}{it[2]}."))
                f, coords, shape = oracle.spconv(feats, coords, 1, shape, w, it[3], it[4], it[5], False)
                feats = np.maximum(f * sc + sh, 0)
    D, H, W = shape
    dense = np.zeros((1, feats.shape[1], D, H, W), dtype=np.float32)
    dense[0, :, coords[:, 1], coords[:, 2], coords[:, 3]] = feats
    return torch.from_numpy(dense.reshape(1, -1, H, W))


def dense_neck_head(sd, x):
    def cbr(x, wkey, bnp, stride=1, pad=1, transpose=False):
        w = sd[wkey]
        y = F.conv_transpose2d(x, w, stride=2) if transpose else F.conv2d(x, w, stride=stride, padding=pad)
        sc, sh = _fold(_sub(sd, bnp))
        return torch.relu(y * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1))
    ups = []
    for b, stride in ((0, 1), (1, 2)):
        x = cbr(x, f"neck.blocks.{b}.1.weight", f"neck.blocks.{b}.2.", stride=stride, pad=1)
        for j in range(5):
            x = cbr(x, f"neck.blocks.{b}.{4 + 3 * j}.weight", f"neck.blocks.{b}.{5 + 3 * j}.")
        if b == 0:
            ups.append(cbr(x, "neck.deblocks.0.0.weight", "neck.deblocks.0.1.", pad=0))
        else:
            ups.append(cbr(x, "neck.deblocks.1.0.weight", "neck.deblocks.1.1.", transpose=True))
    x = torch.cat(ups, dim=1)
    outs = []
    for t in range(6):
        outs.append(F.conv2d(x, sd[f"bbox_head.tasks.{t}.conv_box.weight"], sd[f"bbox_head.tasks.{t}.conv_box.bias"]))
        outs.append(F.conv2d(x, sd[f"bbox_head.tasks.{t}.conv_cls.weight"], sd[f"bbox_head.tasks.{t}.conv_cls.bias"]))
    return x, torch.cat(outs, dim=1)


def run(cfg, sd, infos, feats, sample_frames=2):
    from al3d import synthetic
    from al3d.datasets.anchors import generate_task_anchors
    import random
    threads = usable_cores()
    torch.set_num_threads(threads)
    oracle.set_threads(threads)
    vg = cfg.voxel_generator
    anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    ncls = [len(t["class_names"]) for t in cfg.tasks]
    t_frames = []
    for i in range(sample_frames):
        pts = synthetic.make_point_cloud(1000 + i, nsweeps=10)
        t0 = time.perf_counter()
        vox, c, npts, f = oracle.voxelize(pts, vg.range[:3], vg.voxel_size, [1024, 1024, 40], 10, 60000)
        coords = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
        with torch.no_grad():
            x = sparse_encoder(sd, f, coords, [41, 1024, 1024])
            neck, head = dense_neck_head(sd, x)
            emb = neck.mean(-1).mean(-1)
        hout = head[0].permute(1, 2, 0).reshape(128 * 128, -1).numpy()
        off = 0
        for t, nc in enumerate(ncls):
            na = 2 * nc
            oracle.head_predict(hout, anchors[t], na, nc, off, off + na * 10, 0.1, 0.2, 1000, 83,
                                cfg.test_cfg.post_center_limit_range)
            off += na * 10 + na * nc
        t_frames.append(time.perf_counter() - t0)
    cfgm, run_id, n_boxes = synthetic.pool_arrays(infos)
    n = len(infos)
    xy = np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfgm])
    t0 = time.perf_counter()
    Fm = oracle.l1_map_f32(feats, 2)
    S = oracle.spatial_map(xy, 8)
    D = oracle.combine(n, spatial=S, temporal_id=run_id, feat=Fm, normalize="exp", aggregate="sum")
    random.seed(3407)
    box = np.array([int(b) * 0.04 for b in n_boxes], dtype=np.float64)
    oracle.greedy(D, [], random.choice(range(n)), box, 0.12, 0.0, 600.0)
    t_sel = time.perf_counter() - t0
    t_frame = float(np.mean(t_frames))
    return {"value": round(n / (n * t_frame + t_sel), 4), "unit": "frames/s", "cores": threads,
            "kind": "port",
            "sample": f"{sample_frames} frames of the sweep ({t_frame:.2f} s/frame: oracle C voxelize + "
                      f"sparse encoder + NMS with OpenMP, torch-CPU fp32 neck/head) extrapolated to the "
                      f"{n}-frame pool, plus the full oracle selection ({t_sel:.2f} s)"}
