"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (libal3d_oracle.so).

Imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
bench.py, never by the product package.  ``build()`` compiles the plain-C
restatement with gcc (oracle/Makefile).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libal3d_oracle.so")
_lib = None

c_i64, c_int, c_dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_double
_P = ctypes.c_void_p


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.run(["make", "-C", HERE, "-B", "libal3d_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.al3d_oracle_exp_f64.restype = c_dbl
        _lib.al3d_oracle_exp_f64.argtypes = [c_dbl]
        _lib.al3d_oracle_exp_f32.restype = ctypes.c_float
        _lib.al3d_oracle_exp_f32.argtypes = [ctypes.c_float]
        _lib.al3d_oracle_knn_csr.restype = c_i64
        _lib.al3d_oracle_max_temporal_distance.restype = c_i64
        _lib.al3d_oracle_max_finite.restype = c_dbl
        for f in ("al3d_oracle_greedy_f64", "al3d_oracle_greedy_f32"):
            getattr(_lib, f).restype = c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def exp_f64(x):
    x = _c(x, np.float64)
    out = np.empty_like(x)
    lib().al3d_oracle_exp_f64_array(_p(x), c_i64(x.size), _p(out))
    return out


def ego_xy(car_from_global):
    c = _c(car_from_global, np.float64)
    n = c.shape[0]
    xy = np.empty((n, 2), dtype=np.float64)
    lib().al3d_oracle_ego_xy(_p(c), c_i64(n), _p(xy))
    return xy


def knn(xy, kq):
    xy = _c(xy, np.float64)
    n = xy.shape[0]
    d = np.empty((n, kq), dtype=np.float64)
    i = np.empty((n, kq), dtype=np.int64)
    lib().al3d_oracle_knn(_p(xy), c_i64(n), c_int(kq), _p(d), _p(i))
    return d, i


def knn_csr(knn_d, knn_i):
    n, kq = knn_d.shape
    indptr = np.empty(n + 1, dtype=np.int64)
    indices = np.empty(2 * n * kq, dtype=np.int64)
    weights = np.empty(2 * n * kq, dtype=np.float64)
    nnz = lib().al3d_oracle_knn_csr(_p(_c(knn_d, np.float64)), _p(_c(knn_i, np.int64)),
                                    c_i64(n), c_int(kq), _p(indptr), _p(indices), _p(weights))
    return indptr, indices[:nnz].copy(), weights[:nnz].copy()


def apsp(indptr, indices, weights, row0=0, row1=None):
    n = indptr.shape[0] - 1
    row1 = n if row1 is None else row1
    out = np.empty((row1 - row0, n), dtype=np.float64)
    lib().al3d_oracle_apsp(_p(_c(indptr, np.int64)), _p(_c(indices, np.int64)),
                           _p(_c(weights, np.float64)), c_i64(n), c_i64(row0), c_i64(row1),
                           _p(out))
    return out


def spatial_map(xy, k=8):
    """kNN(k) graph geodesics, f64 [N,N] (spatial_temporal_selector.py:65-107)."""
    d, i = knn(xy, k + 1)
    return apsp(*knn_csr(d, i))


def euclid_map(xy, loc_id):
    xy = _c(xy, np.float64)
    loc_id = _c(loc_id, np.int64)
    n = xy.shape[0]
    out = np.empty((n, n), dtype=np.float64)
    lib().al3d_oracle_euclid_map(_p(xy), _p(loc_id), c_i64(n), _p(out))
    return out


def temporal_map(ids):
    ids = _c(ids, np.int64)
    n = ids.shape[0]
    out = np.empty((n, n), dtype=np.float64)
    lib().al3d_oracle_temporal_map(_p(ids), c_i64(n), _p(out))
    return out


def max_temporal_distance(run_id):
    run_id = _c(run_id, np.int64)
    return int(lib().al3d_oracle_max_temporal_distance(_p(run_id), c_i64(run_id.shape[0])))


def max_finite(a):
    a = _c(a, np.float64)
    return float(lib().al3d_oracle_max_finite(_p(a), c_i64(a.size)))


NORMALIZE = {None: 0, "none": 0, "exp": 1, "linear": 2}
AGGREGATE = {"sum": 0, "min": 1, "max": 2}


def combine(n, spatial=None, temporal_id=None, feat=None, normalize="exp", aggregate="sum",
            lambda_t=1.0, lambda_f=1.0, spatial_scale=1.0, temporal_scale=1.0):
    spatial = _c(spatial, np.float64)
    temporal_id = _c(temporal_id, np.int64)
    feat = _c(feat, np.float32)
    out = np.empty((n, n), dtype=np.float64)
    lib().al3d_oracle_combine(_p(spatial), _p(temporal_id), _p(feat), c_i64(n),
                              c_int(NORMALIZE[normalize]), c_int(AGGREGATE[aggregate]),
                              c_dbl(lambda_t), c_dbl(lambda_f), c_dbl(spatial_scale),
                              c_dbl(temporal_scale), _p(out))
    return out


def combine_rows(n, row0, spatial_rows=None, temporal_id=None, feat_rows=None, normalize="exp",
                 aggregate="sum", lambda_t=1.0, lambda_f=1.0, spatial_scale=1.0, temporal_scale=1.0):
    """Rows [row0, row0+nrows) of ``combine``: ``spatial_rows`` / ``feat_rows`` are [nrows, n] blocks."""
    spatial_rows = _c(spatial_rows, np.float64)
    temporal_id = _c(temporal_id, np.int64)
    feat_rows = _c(feat_rows, np.float32)
    nrows = (spatial_rows if spatial_rows is not None else feat_rows).shape[0]
    out = np.empty((nrows, n), dtype=np.float64)
    lib().al3d_oracle_combine_rows(_p(spatial_rows), _p(temporal_id), _p(feat_rows), c_i64(n), c_i64(row0),
                                   c_i64(nrows), c_int(NORMALIZE[normalize]), c_int(AGGREGATE[aggregate]),
                                   c_dbl(lambda_t), c_dbl(lambda_f), c_dbl(spatial_scale),
                                   c_dbl(temporal_scale), _p(out))
    return out


def l1_map_f32(feats, p=2):
    feats = _c(feats, np.float32)
    n, c = feats.shape
    out = np.empty((n, n), dtype=np.float32)
    lib().al3d_oracle_l1_map_f32(_p(feats), c_i64(n), c_i64(c), c_int(p), _p(out))
    return out


def greedy(D, seeded, first, box_cost, cost_f, start_cost, budget_int, seed_map=None,
           check_seeded=False, cap=None):
    """Returns (status, picks).  status 0 ok, -1 duplicate-pick assertion, -2 capacity."""
    assert D.dtype in (np.float64, np.float32) and D.flags.c_contiguous
    n = D.shape[0]
    seed_map = D if seed_map is None else np.ascontiguousarray(seed_map, dtype=D.dtype)
    seeded = np.ascontiguousarray(seeded, dtype=np.int64)
    box_cost = _c(box_cost, np.float64)
    cap = n + 1 if cap is None else cap
    out = np.empty(cap, dtype=np.int64)
    cnt = c_i64(0)
    fn = lib().al3d_oracle_greedy_f64 if D.dtype == np.float64 else lib().al3d_oracle_greedy_f32
    rc = fn(_p(D), _p(seed_map), c_i64(n), _p(seeded), c_i64(seeded.shape[0]), c_i64(first),
            _p(box_cost), c_dbl(cost_f), c_dbl(start_cost), c_dbl(budget_int),
            c_int(1 if check_seeded else 0), _p(out), c_i64(cap), ctypes.byref(cnt))
    return int(rc), out[:cnt.value].copy()


# ------------------------------------------------------------------ detector side
def voxelize(points, range_min, voxel_size, grid, max_points, max_voxels):
    """One frame: returns (voxels [M,max_points,F], coords_zyx [M,3] i32, num_points [M] i32,
    feat [M,F])."""
    pts = _c(points, np.float32)
    n, f = pts.shape
    rm = _c(range_min, np.float32); vs = _c(voxel_size, np.float32)
    g = _c(grid, np.int32)
    voxels = np.zeros((max_voxels, max_points, f), dtype=np.float32)
    coords = np.zeros((max_voxels, 3), dtype=np.int32)
    npo = np.zeros(max_voxels, dtype=np.int32)
    feat = np.zeros((max_voxels, f), dtype=np.float32)
    fn = lib().al3d_oracle_voxelize
    fn.restype = c_i64
    m = fn(_p(pts), c_i64(n), c_int(f), _p(rm), _p(vs), _p(g), c_int(max_points), c_int(max_voxels),
           _p(voxels), _p(coords), _p(npo), _p(feat))
    return voxels[:m].copy(), coords[:m].copy(), npo[:m].copy(), feat[:m].copy()


def spconv(fin, coords, batch, in_shape, wgt, ksize, stride, pad, subm):
    """One sparse conv layer (no bias/BN).  wgt [kz,ky,kx,Cin,Cout].  Returns
    (fout [n_out,Cout], coords_out [n_out,4], out_shape)."""
    fin = _c(fin, np.float32); coords = _c(coords, np.int32); wgt = _c(wgt, np.float32)
    n, cin = fin.shape
    cout = wgt.shape[-1]
    K = int(np.prod(wgt.shape[:3]))
    cap = n if subm else n * K
    fout = np.zeros((max(cap, 1), cout), dtype=np.float32)
    cout_c = np.zeros((max(cap, 1), 4), dtype=np.int32)
    oshape = np.zeros(3, dtype=np.int32)
    fn = lib().al3d_oracle_spconv
    fn.restype = c_i64
    m = fn(_p(fin), _p(coords), c_i64(n), c_int(batch), _p(_c(in_shape, np.int32)), _p(wgt),
           c_int(cin), c_int(cout), _p(_c(ksize, np.int32)), _p(_c(stride, np.int32)),
           _p(_c(pad, np.int32)), c_int(1 if subm else 0), _p(fout), _p(cout_c), c_i64(cap), _p(oshape))
    assert m >= 0
    return fout[:m].copy(), cout_c[:m].copy(), oshape.tolist()


def box_decode(enc, anchors):
    enc = _c(enc, np.float32).reshape(-1, 10); anc = _c(anchors, np.float32).reshape(-1, 9)
    out = np.empty((enc.shape[0], 9), dtype=np.float32)
    lib().al3d_oracle_box_decode(_p(enc), _p(anc), c_i64(enc.shape[0]), _p(out))
    return out


def rbox_pair(a5, b5):
    """Two boxes (x, y, w, l, r) -> (corners_a [8] = x0..x3 | y0..y3, corners_b [8], intersection area, IoU) as the rotated
    NMS forms them."""
    a, b = _c(a5, np.float32).reshape(5), _c(b5, np.float32).reshape(5)
    ca, cb = np.zeros(8, np.float32), np.zeros(8, np.float32)
    inter, iou = ctypes.c_float(0.0), ctypes.c_float(0.0)
    fn = lib().al3d_oracle_rbox_pair
    fn.restype = None
    fn(_p(a), _p(b), _p(ca), _p(cb), ctypes.byref(inter), ctypes.byref(iou))
    return ca, cb, float(inter.value), float(iou.value)


def rotate_nms(dets_sorted, thresh, post_max):
    """dets [n,5] (x,y,w,l,r) in descending score order -> kept indices."""
    d = _c(dets_sorted, np.float32)
    keep = np.zeros(max(len(d), 1), dtype=np.int32)
    fn = lib().al3d_oracle_rotate_nms
    fn.restype = c_i64
    k = fn(_p(d), c_i64(len(d)), ctypes.c_float(thresh), c_i64(post_max), _p(keep))
    return keep[:k].copy()


def task_detections(boxes, cls_logits, score_thresh, iou_thresh, pre_max, post_max, rng):
    """One (sample, task) after the box decoding: numpy restatement of get_task_detections' class-agnostic branch
    (mg_head.py:981-1063) around the C NMS.  boxes [N, 9] decoded, cls_logits [N, nc] -> (boxes [K, 9], scores, labels)."""
    cls = np.asarray(cls_logits, dtype=np.float32)
    boxes = np.asarray(boxes, dtype=np.float32)
    sc = (1.0 / (1.0 + np.exp(-cls.astype(np.float64)))).astype(np.float32)
    top = sc.max(1); lab = sc.argmax(1)
    idx = np.nonzero(top >= np.float32(score_thresh))[0] if score_thresh > 0.0 else np.arange(len(top))
    order = idx[np.lexsort((idx, -top[idx].astype(np.float64)))][:pre_max]   # score desc, index asc
    cand = boxes[order]
    keep = rotate_nms(cand[:, [0, 1, 3, 4, 8]], iou_thresh, post_max)
    b, s, l = cand[keep], top[order][keep], lab[order][keep]
    m = np.all(b[:, :3] >= np.asarray(rng[:3], np.float32), 1) & np.all(b[:, :3] <= np.asarray(rng[3:], np.float32), 1)
    return b[m], s[m], l[m]


def head_predict(hout, anchors, na, nc, box_off, cls_off, score_thresh, iou_thresh, pre_max,
                 post_max, rng):
    """One (sample, task): the box decoding of the candidates + task_detections.
    hout [HW, CH].  Returns (boxes [K,9], scores [K], labels [K])."""
    hw = hout.shape[0]
    cls = hout[:, cls_off:cls_off + na * nc].reshape(hw * na, nc).astype(np.float32)
    sc = (1.0 / (1.0 + np.exp(-cls.astype(np.float64)))).astype(np.float32)
    top = sc.max(1); lab = sc.argmax(1)
    idx = np.nonzero(top >= np.float32(score_thresh))[0]
    order = idx[np.lexsort((idx, -top[idx].astype(np.float64)))][:pre_max]   # score desc, index asc
    enc = hout[:, box_off:box_off + na * 10].reshape(hw * na, 10)[order]
    boxes = box_decode(enc, anchors[order])
    keep = rotate_nms(boxes[:, [0, 1, 3, 4, 8]], iou_thresh, post_max)
    b, s, l = boxes[keep], top[order][keep], lab[order][keep]
    m = np.all(b[:, :3] >= np.asarray(rng[:3], np.float32), 1) & np.all(b[:, :3] <= np.asarray(rng[3:], np.float32), 1)
    return b[m], s[m], l[m]


def set_threads(n):
    """OpenMP thread count of the oracle (cpu_baseline reports it as `cores`)."""
    lib().al3d_oracle_set_threads(c_int(int(n)))


def argsort_desc(x):
    """torch.argsort(-x) restated: descending, NaN last, ties by ascending index (stable)."""
    x = np.asarray(x, dtype=np.float32)
    return np.argsort(-x, kind="stable")


def frame_entropy(scores):
    """Mean binary entropy of one frame's kept scores (entropy_selector.py:72-75); NaN if empty."""
    s = np.asarray(scores, dtype=np.float32)
    if s.size == 0:
        return np.float32(np.nan)
    h = -s * np.log(s) - (np.float32(1.0) - s) * np.log(np.float32(1.0) - s)
    return h.astype(np.float32).mean(dtype=np.float32)


def merge_sweeps(files, xforms, time_lags, min_distance=1.0):
    """a1 restatement.  files: list of raw [p,5] f32 arrays (file 0 = key frame); xforms: list of
    4x4 f64 or None; time_lags: list of float.  Returns the combined [P,5] f32 cloud."""
    raw = np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=np.float32).reshape(-1, 5) for f in files]))
    off = np.zeros(len(files) + 1, dtype=np.int64)
    off[1:] = np.cumsum([np.asarray(f).reshape(-1, 5).shape[0] for f in files])
    xf = np.zeros((len(files), 12), dtype=np.float64)
    has = np.zeros(len(files), dtype=np.uint8)
    for i, t in enumerate(xforms):
        if t is not None:
            xf[i] = np.asarray(t, dtype=np.float64)[:3, :].reshape(12)
            has[i] = 1
    tl = np.ascontiguousarray(np.asarray(time_lags, dtype=np.float64))
    out = np.empty((raw.shape[0], 5), dtype=np.float32)
    fn = lib().al3d_oracle_merge_sweeps
    fn.restype = c_i64
    n = fn(_p(raw), _p(off), c_int(len(files)), _p(xf), _p(has), _p(tl), ctypes.c_float(min_distance), _p(out))
    return out[:n].copy()


def bev_pool(x, geom, B, lo, dx, nx, depth=None, D=1, fHW=1):
    """base.py:127-163 + bev_pool_cuda.cu:21-44 restated (ascending point order).  x [P,C] (or ctx [BN*fHW,C] with
    depth [P]); geom [P,3]; -> [B, nx0, nx1, nx2*C] float32."""
    x = _c(x, np.float32)
    geom = _c(geom, np.float32)
    depth = _c(depth, np.float32)
    P, C = geom.shape[0], x.shape[1]
    lo, dx, nx = _c(lo, np.float32), _c(dx, np.float32), _c(nx, np.int32)
    out = np.zeros((B, int(nx[0]), int(nx[1]), int(nx[2]) * C), dtype=np.float32)
    lib().al3d_oracle_bev_pool(_p(x), _p(depth), c_int(D), c_int(fHW), _p(geom), c_i64(P), c_int(C), c_int(B),
                               _p(lo), _p(dx), _p(nx), _p(out))
    return out
