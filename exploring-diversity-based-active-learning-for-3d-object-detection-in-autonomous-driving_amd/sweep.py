"""Unlabeled-pool sweep: detector forward over every frame -> BEV embeddings.

Replaces the reference's ``buffer_pred`` loops (det3d/selectors/feature_selector.py:51-85).
New capability: under ``torch.distributed`` each rank sweeps the
frames its sampler hands it and one RCCL all-gather restores the single-process
``[N,512]`` tensor in dataset order on every rank.
"""
import torch


def example_to_device(example, device, non_blocking=False):
    """Host dict -> device (reference det3d/torchie/apis/train.py:84-110)."""
    out = {}
    for k, v in example.items():
        if k in ("anchors", "anchors_mask", "reg_targets", "reg_weights", "labels"):
            out[k] = [t.to(device, non_blocking=non_blocking) for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(device, non_blocking=non_blocking)
        elif k == "calib":
            out[k] = {k1: v1.to(device, non_blocking=non_blocking) for k1, v1 in v.items()}
        else:
            out[k] = v
    return out


def gap_embedding(neck_out):
    """``x.mean(-1).mean(-1)``: mean over W then over H (A.1 quirk 11)."""
    return neck_out.mean(dim=-1).mean(dim=-1)


def gather_in_dataset_order(local_feats, local_index, num_frames, collective_at_world_1=False):
    """All-gather per-rank rows and scatter them back to dataset order.

    ``local_index[r]`` is the dataset index of ``local_feats[r]``.  Ranks may hold
    different row counts (padding rows carry index -1) and sampler wrap-around
    duplicates are harmless: every copy of a frame holds the same embedding.
    ``collective_at_world_1``: run the collectives even in a one-rank group (the one-GPU RCCL
    test: the same calls as at N > 1 through the nccl backend).
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not collective_at_world_1):
        out = torch.empty((num_frames, local_feats.shape[1]), dtype=local_feats.dtype,
                          device=local_feats.device)
        out[local_index] = local_feats
        return out
    world = dist.get_world_size()
    dev = local_feats.device
    cnt = torch.tensor([local_feats.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    rows = int(max(int(c.item()) for c in cnts))
    c = local_feats.shape[1]
    pad_f = torch.zeros((rows, c), dtype=local_feats.dtype, device=dev)
    pad_i = torch.full((rows,), -1, dtype=torch.int64, device=dev)
    pad_f[: local_feats.shape[0]] = local_feats
    pad_i[: local_index.shape[0]] = local_index
    all_f = torch.empty((world * rows, c), dtype=local_feats.dtype, device=dev)
    all_i = torch.empty((world * rows,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_f, pad_f)
    dist.all_gather_into_tensor(all_i, pad_i)
    keep = all_i >= 0
    out = torch.empty((num_frames, c), dtype=local_feats.dtype, device=dev)
    out[all_i[keep]] = all_f[keep]
    return out


import os as _os

# Batch pipelining of the sweep over two HIP streams (AL3D_PIPELINE):
#   "ahead" (default)  the index work of batch i+1 -- voxelization + sparse-conv rulebook, small
#                      latency-bound kernels and every host synchronisation of a batch -- runs on a
#                      side stream while batch i is convolved; the conv phase never waits for the host.
#   "split" (or "1")   the whole sparse half of batch i+1 overlaps the dense half of batch i (+3..5 %
#                      more, but the matrix-core kernels of both halves then contend, which blurs
#                      per-kernel timings).
#   "0" / "off"        one batch at a time on the caller's stream.
_mode = _os.environ.get("AL3D_PIPELINE", "ahead").lower()
PIPELINE = {"1": "split", "split": "split", "0": None, "off": None, "none": None}.get(_mode, "ahead")
_SIDE = {}


# AL3D_SIDE_AFTER_SPARSE=1: in "ahead" mode the side stream starts batch i+1's voxelizer + rulebook when batch i's sparse
# encoder has finished (beside the dense neck) instead of right away (beside the level-0 sparse convolutions, the kernels
# that suffer most from it).  +1.3 % frames/s at the round-3 kernels, but the contention then lands on the dense launches
# (1,559 -> 1,774 us each), i.e. on the kernel whose live roofline the bench line reports: default off
SIDE_AFTER_SPARSE = _os.environ.get("AL3D_SIDE_AFTER_SPARSE", "0") == "1"
# AL3D_SIDE_AFTER_STAGE=k (round 5, dev knob): the side stream's work for batch i+1 is released when stage k of batch i's
# sparse encoder (0 = the input level's six layers) has been executed -- between "at once" (default) and AFTER_SPARSE
SIDE_AFTER_STAGE = int(_os.environ.get("AL3D_SIDE_AFTER_STAGE", "-1"))


# AL3D_NMS_AFTER_SPARSE=1: batch i's decode + NMS launch is held back until batch i+1's sparse encoder is through
# (measured +-0: 2,159 / 2,165 against 2,173 / 2,165 frames/s -- what the sparse layers gain, the dense ones lose; default off)
NMS_AFTER_SPARSE = _os.environ.get("AL3D_NMS_AFTER_SPARSE", "0") == "1"


# AL3D_MAIN_PRIORITY=1: the main (convolution) work runs on a high-priority HIP stream
MAIN_PRIORITY = _os.environ.get("AL3D_MAIN_PRIORITY", "0") == "1"
_HP = {}


def _hp_stream(device):
    key = torch.device(device).index or 0
    if key not in _HP:
        _HP[key] = torch.cuda.Stream(device=device, priority=-1)
    return _HP[key]


# AL3D_SIDE_CUS=n: the side stream may use only n compute units (hipExtStreamCreateWithCUMask); 0 = all
SIDE_CUS = int(_os.environ.get("AL3D_SIDE_CUS", "0"))


def masked_stream(device, n_cus, first_cu=0):
    """A torch stream on ``n_cus`` compute units of ``device`` (0: an ordinary stream)."""
    if n_cus <= 0:
        return torch.cuda.Stream(device=device)
    import ctypes
    from . import lib
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        lib.call("al3d_stream_create_cu_mask", int(n_cus), int(first_cu), ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value, device=device)


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = masked_stream(device, SIDE_CUS)
    return _SIDE[key]


def _tensors_of(obj):
    """Every CUDA tensor reachable from a nest of dicts / lists / tuples / objects with tensor fields."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)
    elif hasattr(obj, "__dict__") and not isinstance(obj, type):
        for v in vars(obj).values():
            if isinstance(v, (torch.Tensor, dict, list, tuple)):
                yield from _tensors_of(v)


def sweep_embeddings(detector, dataloader, device, num_frames=None, with_entropy=False,
                     batch_local_weights=None):
    """Run ``detector(example, return_loss=False, estimate=True)`` over the loader and
    return the ``[N,512]`` embeddings in dataset order (on ``device``).

    ``with_entropy``: also return the ``[N]`` per-frame mean box entropy
    (entropy_selector.py:72-75).  ``batch_local_weights``: UWE's second pass -- a ``[>=B]``
    tensor indexed by the position *inside the batch* that scales each embedding
    (uwe_selector.py:96-99, quirk A.1 #8)."""
    feats, index, ents = [], [], []
    seen = 0
    sampler_idx = None
    sampler = getattr(dataloader, "sampler", None)
    if sampler is not None and hasattr(sampler, "__iter__") and not isinstance(
            sampler, torch.utils.data.SequentialSampler):
        sampler_idx = list(iter(sampler))
    def finish(example, preds, middle):
        nonlocal seen
        emb = gap_embedding(middle[-1])
        if batch_local_weights is not None:
            from . import selector_ops as ops
            emb = ops.scale_rows(emb.contiguous(), batch_local_weights[: emb.shape[0]].contiguous())
        if with_entropy:
            ents.append(preds.frame_entropy() if hasattr(preds, "frame_entropy")
                        else torch.stack([_entropy_of(p["scores"]) for p in preds]))
        feats.append(emb)
        b = emb.shape[0]
        if sampler_idx is not None:
            index.extend(sampler_idx[seen:seen + b])
        else:
            index.extend(range(seen, seen + b))
        seen += b

    on_gpu = torch.device(device).type == "cuda"
    mode = PIPELINE if on_gpu else None
    if mode == "split" and not (hasattr(detector, "sparse_stage") and hasattr(detector, "dense_stage")):
        mode = "ahead"
    if mode == "ahead" and not hasattr(detector, "prepare"):
        mode = None
    with torch.no_grad():
        if mode is None:
            for data_batch in dataloader:
                example = example_to_device(data_batch, device, non_blocking=False)
                preds, middle = detector(example, return_loss=False, estimate=True)
                finish(example, preds, middle)
        else:
            # Software pipeline over batches.  The work of the batch that is due on the caller's
            # stream is enqueued FIRST (it contains no host synchronisation), then the host turns to
            # the side stream, whose row-count read-backs it may wait on while the device keeps
            # working on the main stream.
            caller = torch.cuda.current_stream(device)
            main = caller
            if MAIN_PRIORITY:
                # the convolutions on a HIGH-priority stream: the hardware then hands CU slots to their waves first and the
                # side stream's index work fills what is left, instead of both competing at equal priority
                main = _hp_stream(device)
                main.wait_stream(caller)
            side = _side_stream(device)
            side.wait_stream(main)
            pending = None
            done = []                           # completion events of the batches enqueued on the main stream
            it = iter(dataloader)
            # AL3D_NMS_AFTER_SPARSE: a head that can hold its decode + NMS launch back (MultiGroupHead.defer_nms) releases
            # it when the NEXT batch's sparse encoder is through, like the index work above
            nms_head = getattr(detector, "bbox_head", None)
            if not (NMS_AFTER_SPARSE and mode == "ahead" and SIDE_AFTER_SPARSE and hasattr(detector, "dense_stage")
                    and hasattr(nms_head, "flush_deferred")) or with_entropy:
                nms_head = None
            if nms_head is not None:
                nms_head.defer_nms = True
            try:
              while True:
                  if pending is not None:
                    with torch.cuda.stream(main):
                      example, ahead, ev = pending
                      main.wait_event(ev)
                      if mode == "ahead" and SIDE_AFTER_SPARSE and hasattr(detector, "dense_stage"):
                          # the next batch's index work (random grid traffic) is released only once this batch's
                          # sparse convolutions -- the gather-bound kernels it would slow down -- are through: it
                          # then runs beside the matrix-core-bound neck, which does not notice it
                          x, middle = detector.sparse_stage(example, book=ahead)
                          sparse_done = torch.cuda.Event()
                          sparse_done.record(main)
                          side.wait_event(sparse_done)
                          if nms_head is not None:        # the previous batch's decode + NMS: beside this batch's neck too
                              nms_head.flush_deferred(sparse_done)
                          preds, middle = detector.dense_stage(example, x, middle, estimate=True)
                      elif mode == "ahead":
                          bb = getattr(detector, "backbone", None)
                          if SIDE_AFTER_STAGE >= 0 and bb is not None:
                              def _hook(k, _side=side, _main=main):
                                  if k == SIDE_AFTER_STAGE:
                                      e_ = torch.cuda.Event()
                                      e_.record(_main)
                                      _side.wait_event(e_)
                              bb.stage_hook = _hook
                          try:
                              preds, middle = detector(example, return_loss=False, estimate=True, book=ahead)
                          finally:
                              if bb is not None and SIDE_AFTER_STAGE >= 0:
                                  bb.stage_hook = None
                      else:
                          x, middle = ahead
                          preds, middle = detector.dense_stage(example, x, middle, estimate=True)
                      finish(example, preds, middle)
                      pending = None
                      e = torch.cuda.Event()
                      e.record(main)
                      done.append(e)
                  # bound the run-ahead: the side stream is ~10x faster than the main one and would
                  # otherwise prepare (and keep alive) every remaining batch of the pool at once
                  while len(done) > 1:
                      done.pop(0).synchronize()
                  with torch.cuda.stream(side):
                      try:
                          data_batch = next(it)
                      except StopIteration:
                          break
                      example = example_to_device(data_batch, device, non_blocking=False)
                      ahead = detector.prepare(example) if mode == "ahead" else detector.sparse_stage(example)
                      for t in _tensors_of((example, ahead)):
                          t.record_stream(main)   # produced on the side stream, consumed on the main one
                      ev = torch.cuda.Event()
                      ev.record(side)
                  pending = (example, ahead, ev)
            finally:
                # also on an exception inside the loop (loader error, range check, OOM): a head left deferring would hold
                # every later predict()'s launch -- and its head output -- back for ever
                if nms_head is not None:        # the last batch's launch; later predict() calls launch at once again
                    nms_head.defer_nms = False
                    nms_head.flush_deferred()
            main.wait_stream(side)
            if main is not caller:
                caller.wait_stream(main)
                for t in feats + ents:
                    t.record_stream(caller)
    local = torch.cat(feats, dim=0)
    idx = torch.as_tensor(index, dtype=torch.int64, device=local.device)
    n = num_frames if num_frames is not None else int(idx.max().item()) + 1
    out = gather_in_dataset_order(local, idx, n)
    # checked AFTER the collective, on the gathered tensor: every rank sees the same rows, so all
    # ranks raise together instead of one rank leaving the others inside the all-gather
    _check_range(out)
    if with_entropy:
        e = gather_in_dataset_order(torch.cat(ents).unsqueeze(1), idx, n).squeeze(1)
        return out, e
    return out


def _check_range(local):
    """f16x3 arithmetic carries activations below 65504 only; anything larger turns into inf/NaN in
    the layer's output and from there into the frame's embedding.  One reduction per sweep."""
    from . import detector_ops as D
    from .lib import Al3dError
    if D.MATH == "f16x3" and local.numel() and not bool(torch.isfinite(local).all()):
        bad = int((~torch.isfinite(local).all(dim=1)).sum())
        raise Al3dError(f"{bad} frame embedding(s) are not finite: an activation left the f16x3 range "
                        "(|x| < 65504); rerun with AL3D_MATH=bf16x6 (full fp32 range)")


def _entropy_of(scores):
    """Fallback for detectors that return plain dict lists (not this build's head)."""
    from . import selector_ops as ops
    s = scores.float().contiguous().view(1, 1, -1)
    cnt = torch.tensor([[s.shape[2]]], dtype=torch.int32, device=s.device)
    if s.shape[2] == 0:
        return torch.full((), float("nan"), device=s.device)
    return ops.frame_entropy(s, cnt)[0]


def _weighted_entropy_of(scores, labels, class_weight):
    """Per-frame sum of binary entropy x class weight of the frame's boxes (ppal_selector.py:99-109) for detectors that
    return plain dict lists; a frame without detections gives the empty sum 0."""
    from . import selector_ops as ops
    s = scores.float().contiguous().view(1, 1, -1)
    if s.shape[2] == 0:
        return torch.zeros((), device=s.device)
    lab = labels.to(torch.int32).contiguous().view(1, 1, -1)
    cnt = torch.tensor([[s.shape[2]]], dtype=torch.int32, device=s.device)
    return ops.frame_weighted_entropy(s, lab, cnt, class_weight)[0]
