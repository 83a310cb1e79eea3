"""numpy-in / numpy-out adapter over the product's HIP primitives, shaped like
``oracle/oracle.py`` so ``selector_logic.run_case`` can drive either."""
import numpy as np
import torch

from al3d import selector_ops as ops

DEV = "cuda:0"


def _t(a, dtype):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(DEV)


def knn(xy, kq):
    d, i = ops.knn_2d(_t(xy, np.float64), kq)
    return d.cpu().numpy(), i.cpu().numpy()


def spatial_map(xy, k=8):
    return ops.spatial_map(_t(xy, np.float64), k).cpu().numpy()


def euclid_map(xy, loc_id):
    return ops.euclid_map(_t(xy, np.float64), _t(loc_id, np.int64)).cpu().numpy()


def temporal_map(ids):
    return ops.combine_maps(len(ids), temporal_id=_t(ids, np.int64), normalize="none",
                            aggregate="sum", lambda_t=1.0).cpu().numpy()


def max_finite(a):
    return ops.max_finite(_t(a, np.float64))


def max_temporal_distance(run_id):
    best, count = 0, 0
    for i in range(len(run_id)):
        if i == 0 or run_id[i] == run_id[i - 1]:
            count += 1
        else:
            best = max(best, count)
            count = 1
    return best


def combine(n, spatial=None, temporal_id=None, feat=None, **kw):
    return ops.combine_maps(n, spatial=_t(spatial, np.float64), temporal_id=_t(temporal_id, np.int64),
                            feat=_t(feat, np.float32), device=DEV, **kw).cpu().numpy()


def l1_map_f32(feats, p=2):
    return ops.l1_distance(_t(feats, np.float32), p).cpu().numpy()


def greedy(D, seeded, first, box_cost, cost_f, start_cost, budget_int, seed_map=None,
           check_seeded=False, cap=None):
    Dt = _t(D, D.dtype)
    st = None if seed_map is None else _t(seed_map, D.dtype)
    rc, picks = ops.greedy_kcenter(Dt, [int(s) for s in seeded], first, _t(box_cost, np.float64),
                                   cost_f, start_cost, budget_int, seed_map=st,
                                   check_seeded=check_seeded, cap=cap)
    return rc, np.asarray(picks, dtype=np.int64)


def argsort_desc(x):
    return ops.argsort_desc(_t(x, np.float32)).cpu().numpy()
