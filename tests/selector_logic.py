"""Shared test helper: run one golden selector case through a numeric backend.

``backend`` is any object exposing ``spatial_map(xy, k)``, ``combine(...)``,
``l1_map_f32(feats, p)``, ``temporal_map(ids)``, ``max_finite``,
``max_temporal_distance`` and ``greedy(...)`` with the signatures of
``oracle/oracle.py``.  The control flow restates what each reference selector
class does around those calls (file:line in comments); the product's own
selector classes are tested separately through ``build_selector``.
"""
import json
import random

import numpy as np

SEED = 3407  # tools/active_select.py:76-80


def load_case(path):
    z = np.load(path, allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    fx["buffer"] = json.loads(str(fx["buffer_json"]))
    fx["kwargs"] = json.loads(str(fx["kwargs_json"]))
    fx["logs"] = json.loads(str(fx["logs_json"]))
    fx["cls_name"] = str(fx["cls_name"])
    fx["budget"] = int(fx["budget"])
    return fx


def group_ids(logfiles):
    """TemporalSelector groups by logfile *name* (temporal_selector.py:49-54)."""
    seen = {}
    return np.array([seen.setdefault(str(l), len(seen)) for l in logfiles], dtype=np.int64)


def location_ids(fx):
    log_to_loc = {l["logfile"]: l["location"].split("-")[-1] for l in fx["logs"]}
    seen = {}
    return np.array([seen.setdefault(log_to_loc[str(l)], len(seen)) for l in fx["logfiles"]],
                    dtype=np.int64)


def cost_setup(fx, cost_b=0.04, cost_f=0.12):
    buf = fx["buffer"]
    max_key = str(max(int(k) for k in buf))
    sampled = list(buf[max_key])
    n_boxes = fx["n_boxes"]
    cost = 0
    cost += cost_f * len(sampled)               # base_selector.py:78-86
    for i in sampled:
        cost += int(n_boxes[i]) * cost_b
    budget_int = int(str(fx["budget"] + int(max_key)))
    box_cost = np.array([int(b) * cost_b for b in n_boxes], dtype=np.float64)
    return sampled, float(cost), float(budget_int), box_cost


def run_case(fx, backend, feats=None, maps_out=None):
    """Return (status, full_selected_index_list)."""
    cls = fx["cls_name"]
    kw = fx["kwargs"]
    n = fx["n_boxes"].shape[0]
    cost_b, cost_f = kw.get("cost_b", 0.04), kw.get("cost_f", 0.12)
    sampled, start_cost, budget_int, box_cost = cost_setup(fx, cost_b, cost_f)
    xy = fx["ego_xy"]
    k = kw.get("k", 8)
    random.seed(SEED)
    first = -1
    seed_map = None
    check_seeded = False
    order = "sampled+selected"
    if cls == "SpatialTemporalSelector":
        spatial = backend.spatial_map(xy, k)
        norm, agg = kw.get("normalize", "exp"), kw.get("aggregate", "sum")
        sscale = backend.max_finite(spatial) if norm == "linear" else 1.0
        tscale = float(backend.max_temporal_distance(fx["run_id"])) if norm == "linear" else 1.0
        D = backend.combine(n, spatial=spatial, temporal_id=fx["run_id"], normalize=norm,
                            aggregate=agg, lambda_t=float(kw.get("lambda_t", 1)),
                            spatial_scale=sscale, temporal_scale=tscale)
        if maps_out is not None:
            maps_out.update(raw_spatial=spatial, distance_map=D)
    elif cls == "SpatialSelector":
        D = backend.spatial_map(xy, k)
        check_seeded = True
        if maps_out is not None:
            maps_out.update(raw_spatial=D, distance_map=D)
    elif cls == "TemporalSelector":
        D = backend.temporal_map(group_ids(fx["logfiles"]))
        check_seeded = True
        if maps_out is not None:
            maps_out.update(distance_map=D)
    elif cls == "EuSpatialSelector":
        D = backend.euclid_map(xy, location_ids(fx))
        check_seeded = True
        if maps_out is not None:
            maps_out.update(distance_map=D)
    elif cls == "EntropySelector":
        # entropy_selector.py:88-147: argsort(-entropy) over the unlabeled frames, taken in
        # order under the budget; first pick charged at infos_origin[sorted position] (quirk 7)
        left = [i for i in range(n) if i not in sampled]
        if kw.get("random_sample", False):
            left = random.sample(left, kw["sample_num"])
        ent = np.asarray(fx["entropy"], dtype=np.float32)[left]
        srt = [int(v) for v in backend.argsort_desc(ent)]
        picks = [left[srt[0]]]
        cost = start_cost
        cost += cost_f
        cost += int(fx["n_boxes"][srt[0]]) * cost_b
        sid = 1
        while True:
            idx = left[srt[sid]]
            sid += 1
            assert idx not in picks
            cost += cost_f
            cost += int(fx["n_boxes"][idx]) * cost_b
            if cost > budget_int:
                break
            picks.append(idx)
        return 0, picks + sampled
    elif cls == "PPALSelector":
        # ppal_selector.py:146-239: entropy-ranked pool under delta x budget, then greedy on the
        # L1 map with every row/column outside pool + labelled set to -inf
        left = [i for i in range(n) if i not in sampled]
        ent = np.asarray(fx["entropy"], dtype=np.float32)[left]
        srt = [int(v) for v in backend.argsort_desc(ent)]
        pool = [left[srt[0]]]
        cost = start_cost
        cost += cost_f
        cost += int(fx["n_boxes"][srt[0]]) * cost_b
        limit = budget_int + fx["budget"] * (kw.get("delta", 4) - 1)
        sid = 1
        while True:
            idx = left[srt[sid]]
            sid += 1
            cost += cost_f
            cost += int(fx["n_boxes"][idx]) * cost_b
            if cost > limit:
                break
            pool.append(idx)
        D = np.array(backend.l1_map_f32(feats, kw.get("p", 2)), dtype=np.float32, copy=True)
        keep = np.zeros(n, dtype=bool)
        keep[pool + sampled] = True
        D[~keep] = -np.inf
        D[:, ~keep] = -np.inf
        order = "selected+sampled"
        if maps_out is not None:
            maps_out.update(distance_map=D)
    elif cls in ("FeatureSelector", "BadgeSelector", "UWESelector"):
        D = backend.l1_map_f32(feats, kw.get("p", 2))
        order = "selected+sampled"
        if maps_out is not None:
            maps_out.update(distance_map=D)
    elif cls == "SpatialTemporalFeatureSelector":
        spatial = backend.spatial_map(xy, k)
        F = backend.l1_map_f32(feats, kw.get("p", 2))
        D = backend.combine(n, spatial=spatial, temporal_id=fx["run_id"], feat=F,
                            normalize="exp", aggregate="sum",
                            lambda_t=float(kw.get("lambda_t", 1)),
                            lambda_f=float(kw.get("lambda_f", 1)))
        if maps_out is not None:
            maps_out.update(raw_spatial=spatial, feature_raw=F, distance_map=D)
    elif cls == "SpatialFeatureSelector":
        spatial = backend.spatial_map(xy, k)
        F = backend.l1_map_f32(feats, kw.get("p", 2))
        seed_map = backend.combine(n, spatial=spatial, normalize="exp", aggregate="sum")
        D = backend.combine(n, spatial=spatial, feat=F, normalize="exp",
                            aggregate=kw.get("aggregate", "sum"), lambda_f=1.0)
        check_seeded = True
        if maps_out is not None:
            maps_out.update(raw_spatial=spatial, distance_map=D)
    else:
        raise NotImplementedError(cls)
    if len(sampled) == 0:
        first = random.choice(range(n))
    rc, picks = backend.greedy(D, sampled, first, box_cost, cost_f, start_cost, budget_int,
                               seed_map=seed_map, check_seeded=check_seeded)
    picks = [int(p) for p in picks]
    full = sampled + picks if order == "sampled+selected" else picks + sampled
    return rc, full


def ulp_diff_f64(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    # sign-magnitude -> monotone integer line, then exact int64 distance
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    d = np.abs(ia - ib).astype(np.float64)
    d[same_inf] = 0
    return d


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
