"""Device-side selector primitives: torch tensors in, HIP kernels underneath.

PyTorch is plumbing here (device memory + the current stream); every number is
produced by libal3d_hip.so.  All functions require CUDA(ROCm) tensors and raise
if handed CPU ones -- there is no host fallback.
"""
import torch

from . import lib

NORMALIZE = {None: 0, "none": 0, "exp": 1, "linear": 2}
SHARD_MIN_ROWS = 4096          # pools below this are cheaper to map on every rank than to gather
AGGREGATE = {"sum": 0, "min": 1, "max": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise lib.Al3dError(f"{name}: expected a device tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise lib.Al3dError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def l1_distance(feats, p=2, shard=True):
    """[N,C] f32 embeddings -> [N,N] f32 L1 map (feature_selector.py:87-109).

    Under torch.distributed (and ``shard``) this is a COLLECTIVE: every rank computes one block of
    rows and the blocks are all-gathered, so every rank must call it.  ``shard=False`` computes the
    whole map locally (what a rank-0-only checker needs)."""
    import torch.distributed as dist
    feats = _dev(feats, torch.float32, "feats")
    n, c = feats.shape
    if (not shard or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1
            or n < SHARD_MIN_ROWS):
        out = torch.empty((n, n), dtype=torch.float32, device=feats.device)
        lib.call("al3d_l1_distance_f32", _ptr(feats), n, c, int(p), _ptr(out), _stream())
        return out
    # N>1: every rank holds all embeddings (they were all-gathered); each computes a block of rows
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n + world - 1) // world
    r0 = min(n, rank * per)
    nrows = min(n, r0 + per) - r0
    block = torch.empty((per, n), dtype=torch.float32, device=feats.device)
    lib.call("al3d_l1_distance_rows_f32", _ptr(feats), n, c, int(p), r0, nrows, _ptr(block), _stream())
    full = torch.empty((world * per, n), dtype=torch.float32, device=feats.device)
    dist.all_gather_into_tensor(full, block)
    return full[:n]


def combine_maps(n, spatial=None, temporal_id=None, feat=None, normalize="exp", aggregate="sum",
                 lambda_t=1.0, lambda_f=1.0, spatial_scale=1.0, temporal_scale=1.0, device=None):
    spatial = _dev(spatial, torch.float64, "spatial")
    temporal_id = _dev(temporal_id, torch.int64, "temporal_id")
    feat = _dev(feat, torch.float32, "feat")
    for t in (spatial, temporal_id, feat):
        if t is not None:
            device = t.device
    out = torch.empty((n, n), dtype=torch.float64, device=device)
    lib.call("al3d_combine_maps_f64", _ptr(spatial), _ptr(temporal_id), _ptr(feat), n,
             NORMALIZE[normalize], AGGREGATE[aggregate], float(lambda_t), float(lambda_f),
             float(spatial_scale), float(temporal_scale), _ptr(out), _stream())
    return out


def euclid_map(xy, loc_id):
    """EuSpatialSelector map (euclidean_spatial_selector.py:95-106)."""
    xy = _dev(xy, torch.float64, "xy")
    loc_id = _dev(loc_id, torch.int64, "loc_id")
    n = xy.shape[0]
    out = torch.empty((n, n), dtype=torch.float64, device=xy.device)
    lib.call("al3d_euclid_map_f64", _ptr(xy), _ptr(loc_id), n, _ptr(out), _stream())
    return out


def max_finite(a):
    a = _dev(a, torch.float64, "a")
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    lib.call("al3d_max_finite_f64", _ptr(a), a.numel(), _ptr(out), _stream())
    return float(out.item())


def knn_2d(xy, kq):
    xy = _dev(xy, torch.float64, "xy")
    n = xy.shape[0]
    d = torch.empty((n, kq), dtype=torch.float64, device=xy.device)
    i = torch.empty((n, kq), dtype=torch.int64, device=xy.device)
    lib.call("al3d_knn_2d_f64", _ptr(xy), n, int(kq), _ptr(d), _ptr(i), _stream())
    return d, i


def apsp_knn(knn_d, knn_i, row0=0, nrows=None):
    """Geodesic rows [row0, row0+nrows) of the kNN graph, f64 [nrows, n]."""
    knn_d = _dev(knn_d, torch.float64, "knn_d")
    knn_i = _dev(knn_i, torch.int64, "knn_i")
    n, kq = knn_d.shape
    nrows = n - row0 if nrows is None else nrows
    out = torch.empty((nrows, n), dtype=torch.float64, device=knn_d.device)
    ws = torch.empty(max(1, lib.load().al3d_apsp_workspace_bytes(n, kq)), dtype=torch.uint8,
                     device=knn_d.device)
    lib.call("al3d_apsp_knn_rows_f64", _ptr(knn_d), _ptr(knn_i), n, kq, int(row0), int(nrows), _ptr(out),
             _ptr(ws), _stream())
    return out


def spatial_map(xy, k=8):
    """kNN(k)-graph geodesic map, f64 [N,N] (spatial_temporal_selector.py:92-104).

    Under torch.distributed the N source rows are sharded over the ranks (each computes a
    contiguous block) and all-gathered -- the sweeps are independent per source.  That makes this a
    COLLECTIVE (every rank must call it); ``knn_2d`` + ``apsp_knn`` is the local form."""
    import torch.distributed as dist
    d, i = knn_2d(xy, k + 1)
    n = d.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or n < SHARD_MIN_ROWS:
        return apsp_knn(d, i)
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n + world - 1) // world
    r0 = min(n, rank * per)
    rows = apsp_knn(d, i, r0, min(n, r0 + per) - r0)
    pad = torch.empty((per, n), dtype=torch.float64, device=d.device)
    pad[: rows.shape[0]] = rows
    full = torch.empty((world * per, n), dtype=torch.float64, device=d.device)
    dist.all_gather_into_tensor(full, pad)
    return full[:n]


def frame_entropy(scores, counts):
    """scores [B,nt,post] f32, counts [B,nt] i32 -> [B] mean binary entropy per frame."""
    scores = _dev(scores, torch.float32, "scores")
    counts = _dev(counts, torch.int32, "counts")
    B, nt, post = scores.shape
    out = torch.empty((B,), dtype=torch.float32, device=scores.device)
    lib.call("al3d_frame_entropy_f32", _ptr(scores), _ptr(counts), B, nt, post, _ptr(out), _stream())
    return out


def frame_weighted_entropy(scores, labels, counts, class_weight):
    scores = _dev(scores, torch.float32, "scores")
    labels = _dev(labels, torch.int32, "labels")
    counts = _dev(counts, torch.int32, "counts")
    cw = _dev(class_weight, torch.float32, "class_weight")
    B, nt, post = scores.shape
    out = torch.empty((B,), dtype=torch.float32, device=scores.device)
    lib.call("al3d_frame_weighted_entropy_f32", _ptr(scores), _ptr(labels), _ptr(counts), B, nt, post,
             _ptr(cw), cw.numel(), _ptr(out), _stream())
    return out


def mask_map_(D, keep):
    """In place: D[i,j] = -inf unless keep[i] and keep[j] (keep: uint8 [n])."""
    D = _dev(D, torch.float32, "D")
    keep = _dev(keep, torch.uint8, "keep")
    lib.call("al3d_mask_map_f32", _ptr(D), D.shape[0], _ptr(keep), _stream())
    return D


def scale_rows(feats, w, widx=None):
    feats = _dev(feats, torch.float32, "feats")
    w = _dev(w, torch.float32, "w")
    widx = _dev(widx, torch.int64, "widx")
    n, c = feats.shape
    out = torch.empty_like(feats)
    lib.call("al3d_scale_rows_f32", _ptr(feats), _ptr(w), _ptr(widx), n, c, _ptr(out), _stream())
    return out


def minmax_norm(x):
    x = _dev(x, torch.float32, "x")
    out = torch.empty_like(x)
    lib.call("al3d_minmax_norm_f32", _ptr(x), x.numel(), _ptr(out), _stream())
    return out


def argsort_desc(x):
    """torch.argsort(-x) semantics: descending, NaN last, ties by ascending index."""
    x = _dev(x, torch.float32, "x")
    n = x.numel()
    out = torch.empty((n,), dtype=torch.int64, device=x.device)
    ws = torch.empty(max(8, lib.load().al3d_argsort_workspace_bytes(n)), dtype=torch.uint8, device=x.device)
    lib.call("al3d_argsort_desc_f32", _ptr(x), n, _ptr(out), _ptr(ws), _stream())
    return out


def greedy_kcenter(D, seeded, first, box_cost, cost_f, start_cost, budget_int, seed_map=None,
                   check_seeded=False, cap=None):
    """Run the whole pick loop on device.  Returns (status, picks[list[int]])."""
    if D.dtype not in (torch.float64, torch.float32):
        raise lib.Al3dError(f"greedy_kcenter: unsupported map dtype {D.dtype}")
    D = _dev(D, D.dtype, "D")
    seed_map = D if seed_map is None else _dev(seed_map, D.dtype, "seed_map")
    n = D.shape[0]
    dev = D.device
    seeded_t = torch.as_tensor(list(seeded), dtype=torch.int64, device=dev)
    if seeded_t.numel() and (int(seeded_t.min()) < 0 or int(seeded_t.max()) >= n):
        raise lib.Al3dError("greedy_kcenter: seeded index outside the pool")
    box_cost = _dev(box_cost, torch.float64, "box_cost")
    cap = n + 1 if cap is None else int(cap)
    out_idx = torch.empty(cap, dtype=torch.int64, device=dev)
    meta = torch.zeros(2, dtype=torch.int64, device=dev)
    esz = D.element_size()
    ws = torch.empty(lib.load().al3d_greedy_workspace_bytes(n, esz), dtype=torch.uint8, device=dev)
    fn = "al3d_greedy_kcenter_f64" if D.dtype == torch.float64 else "al3d_greedy_kcenter_f32"
    lib.call(fn, _ptr(D), _ptr(seed_map), n, _ptr(seeded_t) if seeded_t.numel() else None,
             int(seeded_t.numel()), int(first), _ptr(box_cost), float(cost_f), float(start_cost),
             float(budget_int), 1 if check_seeded else 0, _ptr(out_idx), cap, _ptr(meta), _ptr(ws),
             _stream())
    cnt, status = meta.tolist()
    return int(status), out_idx[:cnt].tolist()
