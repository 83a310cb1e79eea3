#!/usr/bin/env python3
"""Selection CLI -- same flags and flow as the reference's tools/active_select.py:22-163.

    python tools/active_select.py --config examples/active/cbgs_spatial_temporal.py --budget 600
    torchrun --nproc-per-node 8 tools/active_select.py --config … --pred       # sharded sweep

Seeds (3407), the empty-buffer bootstrap, ``build_detector`` / ``build_selector`` /
``select_samples(local_rank=…)`` / ``dump_file()`` follow the reference.  Differences: the
detector is only built when the selector sweeps (``--pred``) -- the reference builds it and calls
``.cuda()`` even for metadata-only selectors; the pool is staged in HBM
(``--synthetic-scenes`` generates one, otherwise ``cfg.selector.infos_origin`` + ``cfg.data_root``
are read with the reference's loading rules: det3d infos for the CBGS / lidar-only models, mmdet3d-format infos with
``cams`` for the BEVFusion camera+lidar detector); under torch.distributed every rank sweeps a
contiguous shard and the embeddings are all-gathered.
"""
import argparse
import logging
import os
import pickle
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="Active-learning sample selection")
    p.add_argument("--config", default="examples/active/cbgs_spatial_temporal.py")
    p.add_argument("--work_dir")
    p.add_argument("--checkpoint", default=None)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--budget", type=int, default=None)
    p.add_argument("--pred", action="store_true", help="sweep the pool and save the embeddings")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="pytorch")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--synthetic-scenes", type=int, default=0,
                   help="generate a synthetic pool of this many scenes (writes infos/log files)")
    p.add_argument("--batch", type=int, default=8)
    args = p.parse_args()
    os.environ.setdefault("LOCAL_RANK", str(args.local_rank))
    return args


def main():
    torch.manual_seed(3407)
    np.random.seed(3407)
    random.seed(3407)
    args = parse_args()
    from al3d import synthetic
    from al3d.selectors import build_selector
    from al3d.utils import Config, fileio

    cfg = Config.fromfile(args.config)
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    if args.budget is not None:
        cfg.selector.budget = args.budget
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if distributed:
        torch.cuda.set_device(local_rank)
        # RCCL ("nccl" on ROCm), one GPU per rank; AL3D_DIST_BACKEND=gloo is the one-GPU rehearsal knob
        torch.distributed.init_process_group(backend=os.environ.get("AL3D_DIST_BACKEND", "nccl"),
                                             init_method="env://")
    rank = torch.distributed.get_rank() if distributed else 0
    world = torch.distributed.get_world_size() if distributed else 1
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.ERROR)
    logger = logging.getLogger("active_select")
    logger.info("Distributed testing: %s", distributed)

    sel_cfg = cfg.selector
    if args.synthetic_scenes:
        infos, logs = synthetic.make_pool(args.synthetic_scenes, seed=0)
        os.makedirs(os.path.dirname(sel_cfg.infos_origin) or ".", exist_ok=True)
        if rank == 0:
            with open(sel_cfg.infos_origin, "wb") as f:
                pickle.dump(infos, f)
            fileio.dump(logs, os.path.join(os.path.dirname(sel_cfg.infos_origin), "log.json"))
        if distributed:
            torch.distributed.barrier()
        import inspect
        from al3d.selectors import SELECTORS
        accepted = inspect.signature(SELECTORS.get(sel_cfg.type).__init__).parameters
        if "logs_file" in accepted:                     # the map selectors; the uncertainty selectors take no scene logs
            sel_cfg["logs_file"] = os.path.join(os.path.dirname(sel_cfg.infos_origin), "log.json")
        if "distance_store_file" in accepted:
            sel_cfg.setdefault("distance_store_file", None)
        # the selectors' own defaults for their caches are the reference authors' absolute paths
        # (spatial_temporal_feature_selector.py:24, feature_selector.py:20, ...): a synthetic run keeps them next
        # to the buffer file instead
        for key, name in (("buffer_path", "feature_pred.pt"), ("weighted_feat_path", "weighted_feature_pred.pt")):
            if key in accepted:
                sel_cfg.setdefault(key, os.path.join(os.path.dirname(sel_cfg.buffer_file) or ".", name))
    os.makedirs(os.path.dirname(sel_cfg.buffer_file) or ".", exist_ok=True)
    if not os.path.exists(sel_cfg.buffer_file):
        # first round: empty buffer (reference init_sample_dataset, active_select.py:68-71)
        if rank == 0:
            fileio.dump({"0": []}, sel_cfg.buffer_file, indent=4)
        logger.info("init a empty buffer, and save as %s", sel_cfg.buffer_file)
        return

    model = loader = None
    if args.pred:
        from al3d.datasets import DeviceSweepLoader, PoolFrames, generate_task_anchors
        from al3d.models import build_detector
        dev = torch.device("cuda", local_rank)
        model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        if args.checkpoint:
            sd = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
            sd = sd.get("state_dict", sd)
            sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
            missing, unexpected = model.load_state_dict(sd, strict=False)
            logger.info("checkpoint loaded (missing %d, unexpected %d)", len(missing), len(unexpected))
        else:
            synthetic.seeded_init_(model, seed=0)
        model = model.to(dev).eval()
        infos = fileio.load(sel_cfg.infos_origin)
        n = len(infos)
        per = (n + world - 1) // world
        mine = list(range(rank * per, min(n, (rank + 1) * per)))
        # embedding-only models (bbox_head=None, e.g. the BEVFusion lidar branch) and anchor-free heads need no anchors
        head_cfg = cfg.model.get("bbox_head")
        anchors = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128]) \
            if head_cfg is not None and head_cfg.get("type") == "MultiGroupHead" else None      # TransFusionHead is anchor-free
        if args.synthetic_scenes and cfg.model.get("type") == "BEVFusion":
            # camera+lidar: the lidar batch plus synthetic six-camera images and one calibration rig (configs[4])
            from al3d.datasets import CameraLidarSweepLoader
            pool = PoolFrames.from_synthetic(len(mine), dev, seed=1000 + rank)
            cam = cfg.get("camera", {})
            loader = CameraLidarSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=args.batch, device=dev,
                                            image_size=tuple(cam.get("image_size", (256, 704))),
                                            num_cameras=int(cam.get("num_cameras", 6)))
        elif cfg.model.get("type") == "BEVFusion":
            # camera+lidar from an mmdet3d-format pool (configs[4]): key frame + sweeps through the native reader and the
            # BEVFusion merge rule, six camera frames per sample decoded on host threads and resized / cropped / normalised
            # on the device (the reference's LoadMultiViewImageFromFiles / LoadPointsFromMultiSweeps / ImageAug3D /
            # ImageNormalize test pipeline)
            from al3d.datasets import CameraLidarFileLoader
            cam = cfg.get("camera", {})
            loader = CameraLidarFileLoader([infos[i] for i in mine], cfg.voxel_generator, anchors, batch_size=args.batch,
                                           device=dev, sweeps_num=int(cam.get("sweeps_num", 9)), root=cfg.data_root,
                                           threads=int(os.environ.get("AL3D_READER_THREADS", "8")),
                                           image_size=tuple(cam.get("image_size", (256, 704))),
                                           resize_lim=tuple(cam.get("resize_test", (0.48, 0.48))))
        elif args.synthetic_scenes:
            pool = PoolFrames.from_synthetic(len(mine), dev, seed=1000 + rank)
            loader = DeviceSweepLoader(pool, cfg.voxel_generator, anchors, batch_size=args.batch, device=dev)
        else:
            # real files: streamed by the native reader pool one batch ahead of the detector (the reference's
            # 8 DataLoader workers, build_loader.py:23-59), never staged as a whole pool
            from al3d.datasets import FileSweepLoader
            loader = FileSweepLoader([infos[i] for i in mine], cfg.voxel_generator, anchors, batch_size=args.batch,
                                     device=dev, nsweeps=cfg.nsweeps, root=cfg.data_root,
                                     threads=int(os.environ.get("AL3D_READER_THREADS", "8")))
        loader.sampler = mine

    sel_cfg.update({"detector": model, "dataloader": loader, "logger": logger, "pred": args.pred})
    selector = build_selector(dict(sel_cfg))
    logger.info("begin selection")
    selector.select_samples(local_rank=local_rank)
    selector.dump_file()


if __name__ == "__main__":
    main()
