"""GPU suite: the HIP selector path against the CPU oracle and the reference's golden
vectors.  Integer/index work and the f64 maps must be bit-exact with the oracle."""
import glob
import hashlib
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch

import selector_logic as L
from al3d import synthetic

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "selector_*.npz")))
IDS = [os.path.basename(p)[9:-4] for p in CASES]


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import hip_backend
    return hip_backend


def _feats(fx):
    if "feats_seed" not in fx:
        return None
    return synthetic.make_embeddings(fx["n_boxes"].shape[0], seed=int(fx["feats_seed"]),
                                     scale=float(fx["feats_scale"]))


# ---------------------------------------------------------------- primitives vs oracle
@pytest.mark.parametrize("n,kq", [(1, 9), (5, 9), (257, 9), (1000, 5), (2560, 9)])
def test_knn_bit_exact(hip, oracle, n, kq):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 500, size=(n, 2))
    if n > 10:
        xy[3] = xy[1]  # coincident points
    d0, i0 = oracle.knn(xy, kq)
    d1, i1 = hip.knn(xy, kq)
    assert np.array_equal(i0, i1)
    assert np.array_equal(d0.view(np.int64), d1.view(np.int64))


@pytest.mark.parametrize("n,k,scale", [(64, 8, 50.0), (700, 8, 300.0), (1500, 4, 200.0)])
def test_apsp_bit_exact(hip, oracle, n, k, scale):
    rng = np.random.default_rng(n + k)
    xy = rng.uniform(0, scale, size=(n, 2))
    xy[5] = xy[2]
    ref = oracle.spatial_map(xy, k)
    got = hip.spatial_map(xy, k)
    assert np.array_equal(np.isinf(ref), np.isinf(got))
    assert np.array_equal(ref.view(np.int64), got.view(np.int64))
    fin = np.isfinite(got)                      # undirected graph => symmetric up to path-sum order
    assert np.array_equal(fin, fin.T) and np.allclose(got[fin], got.T[fin], rtol=1e-13)


@pytest.mark.parametrize("norm", ["exp", "linear", "none"])
@pytest.mark.parametrize("agg", ["sum", "min", "max"])
def test_combine_bit_exact(hip, oracle, norm, agg):
    n = 333
    rng = np.random.default_rng(7)
    S = rng.uniform(0, 60, size=(n, n))
    S[rng.uniform(size=(n, n)) < 0.2] = np.inf
    np.fill_diagonal(S, 0.0)
    F = rng.uniform(0, 8, size=(n, n)).astype(np.float32)
    ids = np.sort(rng.integers(0, 9, size=n)).astype(np.int64)
    for kw in (dict(spatial=S, temporal_id=ids), dict(spatial=S, temporal_id=ids, feat=F),
               dict(spatial=S, feat=F), dict(temporal_id=ids), dict(spatial=S)):
        args = dict(normalize=norm, aggregate=agg, lambda_t=0.7, lambda_f=1.3,
                    spatial_scale=59.0, temporal_scale=41.0, **kw)
        ref = oracle.combine(n, **args)
        got = hip.combine(n, **args)
        assert np.array_equal(ref.view(np.int64), got.view(np.int64)), (norm, agg, list(kw))


@pytest.mark.parametrize("n,c,p", [(1, 512, 2), (63, 512, 1), (130, 512, 2), (200, 100, 2), (257, 7, 1)])
def test_l1_map_bit_exact(hip, oracle, n, c, p):
    rng = np.random.default_rng(c)
    f = np.abs(rng.normal(size=(n, c))).astype(np.float32)
    f[0, : min(c, 3)] = [1e-30, 3e25, 0.0][: min(c, 3)]   # exercise the sqrt(d*d) edge range
    ref = oracle.l1_map_f32(f, p)
    got = hip.l1_map_f32(f, p)
    assert np.array_equal(ref.view(np.int32), got.view(np.int32))


def test_max_finite(hip, oracle):
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 1e3, size=100003)
    a[::7] = np.inf
    assert hip.max_finite(a) == oracle.max_finite(a)
    assert hip.max_finite(np.full(10, np.inf)) == -np.inf


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,seeded", [(50, 0), (1500, 0), (1500, 7), (4099, 3)])
def test_greedy_bit_exact(hip, oracle, dtype, n, seeded):
    rng = np.random.default_rng(n + seeded)
    # quantised values => plenty of exact ties for the first-index rule
    D = (rng.integers(0, 40, size=(n, n)) / 8.0).astype(dtype)
    D = np.minimum(D, D.T)
    np.fill_diagonal(D, 0)
    box = rng.integers(0, 70, size=n) * 0.04
    seeds = rng.choice(n, size=seeded, replace=False).tolist()
    first = int(rng.integers(0, n)) if not seeds else -1
    for budget in (3.0, 60.0, 300.0):
        r0, p0 = oracle.greedy(D, seeds, first, box, 0.12, 0.37, budget)
        r1, p1 = hip.greedy(D, seeds, first, box, 0.12, 0.37, budget)
        assert r0 == r1 and p0.tolist() == p1.tolist()


def test_greedy_duplicate_and_capacity(hip, oracle):
    n = 16
    D = np.ones((n, n))
    np.fill_diagonal(D, 0.0)
    box = np.zeros(n)
    rc, picks = hip.greedy(D, [], 5, box, 0.12, 0.0, 1.0)
    assert rc == 0 and picks.tolist() == [5, 0, 1, 2, 3, 4, 6, 7]
    rc, picks = hip.greedy(D, [], 0, box, 0.12, 0.0, 100.0)
    assert rc == -1 and len(picks) == n            # reference assert would fire (A.1 #13)
    rc, picks = hip.greedy(D, [], 0, box, 0.12, 0.0, 100.0, cap=4)
    assert rc == -2 and len(picks) == 4
    rc, picks = hip.greedy(D, [3, 9], -1, box, 0.12, 0.0, 100.0, check_seeded=True)
    ro, po = oracle.greedy(D, [3, 9], -1, box, 0.12, 0.0, 100.0, check_seeded=True)
    assert (rc, picks.tolist()) == (ro, po.tolist())


def test_cpu_tensors_are_rejected():
    from al3d import lib, selector_ops as ops
    with pytest.raises(lib.Al3dError):
        ops.l1_distance(torch.zeros(4, 4))


# ---------------------------------------------------------------- golden cases, HIP numerics
@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_hip_matches_reference_and_oracle(hip, oracle, path):
    fx = L.load_case(path)
    feats = _feats(fx)
    m_hip, m_orc = {}, {}
    rc, full = L.run_case(fx, hip, feats=feats, maps_out=m_hip)
    rc_o, full_o = L.run_case(fx, oracle, feats=feats, maps_out=m_orc)
    assert rc == rc_o and full == full_o
    if str(fx["error"]) == "AssertionError":
        assert rc == -1
        return
    assert full == fx["selected"].tolist()          # identical selected-index buffer
    for k in m_orc:                                 # every intermediate map, bit for bit
        a, b = np.ascontiguousarray(m_orc[k]), np.ascontiguousarray(m_hip[k])
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), k


# ---------------------------------------------------------------- product classes end to end
def _write_case(tmp, fx):
    n = fx["n_boxes"].shape[0]
    infos = []
    for i in range(n):
        lf = str(fx["logfiles"][i])
        infos.append({
            "car_from_global": fx["car_from_global"][i],
            "cam_front_path": f"samples/CAM_FRONT/{lf}__CAM_FRONT__{i}.jpg",
            "gt_names": np.array(["car"] * int(fx["n_boxes"][i]), dtype="<U32"),
        })
    ip = os.path.join(tmp, "infos.pkl")
    with open(ip, "wb") as f:
        pickle.dump(infos, f)
    lp = os.path.join(tmp, "log.json")
    with open(lp, "w") as f:
        json.dump(fx["logs"], f)
    bp = os.path.join(tmp, "buffer.json")
    with open(bp, "w") as f:
        json.dump(fx["buffer"], f)
    return ip, lp, bp


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_build_selector_end_to_end(path, tmp_path):
    """Same call sequence as tools/active_select.py:152-163 of the reference."""
    from al3d.selectors import build_selector
    fx = L.load_case(path)
    tmp = str(tmp_path)
    ip, lp, bp = _write_case(tmp, fx)
    cfg = dict(type=fx["cls_name"], budget=fx["budget"], buffer_file=bp, infos_origin=ip, pred=False)
    cls = fx["cls_name"]
    if "Feature" in cls:
        fp = os.path.join(tmp, "feats.pt")
        torch.save(torch.from_numpy(_feats(fx)), fp)
        cfg["buffer_path"] = fp
    if cls in ("BadgeSelector", "UWESelector"):
        fp = os.path.join(tmp, "wfeats.pt")
        torch.save(torch.from_numpy(_feats(fx)), fp)
        cfg["weighted_feat_path"] = fp
    if cls == "PPALSelector":
        fp = os.path.join(tmp, "pfeats.pt")
        torch.save(torch.from_numpy(_feats(fx)), fp)
        ep = os.path.join(tmp, "pent.pt")
        torch.save(torch.from_numpy(fx["entropy"]), ep)
        cfg.update(feat_path=fp, ent_path=ep)
    if cls == "EntropySelector":
        fp = os.path.join(tmp, "entropy.pt")
        torch.save(torch.from_numpy(fx["entropy"]), fp)
        cfg["buffer_path"] = fp
    if cls not in ("FeatureSelector", "TemporalSelector", "RandomSelector", "EntropySelector",
                   "BadgeSelector", "UWESelector", "PPALSelector"):
        cfg["logs_file"] = lp
    if cls not in ("TemporalSelector", "RandomSelector", "EntropySelector"):
        cfg["distance_store_file"] = os.path.join(tmp, "dist.npy")
    cfg.update(fx["kwargs"])
    random.seed(L.SEED)
    sel = build_selector(cfg)
    if str(fx["error"]) == "AssertionError":
        with pytest.raises(AssertionError):
            sel.select_samples(local_rank=0)
        return
    sel.select_samples(local_rank=0)
    key = str(fx["current_budget"])
    assert sel.current_budget == key
    assert sel.get_selected_samples()[key] == fx["selected"].tolist()
    sel.dump_file()
    assert json.load(open(bp))[key] == fx["selected"].tolist()
    out_infos = pickle.load(open(ip.replace(".pkl", f"_{key}.pkl"), "rb"))
    assert len(out_infos) == len(fx["selected"])
    if "distance_store_file" in cfg and cls != "FeatureSelector" and "ref_raw_spatial_map" in fx:
        cached = np.load(cfg["distance_store_file"])
        assert np.array_equal(cached.view(np.int64), fx["ref_raw_spatial_map"].view(np.int64))


# ---------------------------------------------------------------- full-size properties
def test_full_size_pool_properties(hip, oracle):
    """BASELINE.json configs[1] size (64 scenes, N=2560): properties that do not need the
    oracle at full size, plus the oracle itself (it finishes in seconds at this N)."""
    infos, logs = synthetic.make_pool(64, seed=0)
    cfg, run_id, n_boxes = synthetic.pool_arrays(infos)
    xy = np.stack([(-(c[:3, 3].T @ c[:3, :3]))[:2] for c in cfg])
    n = len(infos)
    S = hip.spatial_map(xy, 8)
    assert np.all(np.diag(S) == 0)
    assert np.array_equal(S.view(np.int64), oracle.spatial_map(xy, 8).view(np.int64))
    D = hip.combine(n, spatial=S, temporal_id=run_id, normalize="exp", aggregate="sum", lambda_t=1.0)
    assert D.min() >= 0 and D.max() <= 2.0 and np.allclose(D, D.T, rtol=1e-13)
    box = n_boxes * 0.04
    rc, picks = hip.greedy(D, [], 232, box, 0.12, 0.0, 600.0)
    assert rc == 0 and len(set(picks.tolist())) == len(picks)
    # k-center invariant: every pick was the farthest point from the set picked before it
    fps = D[picks[0]].copy()
    for p in picks[1:]:
        assert fps[p] == fps.max() and p == int(np.argmax(fps))
        fps = np.minimum(fps, D[p])
    cost = 0.0
    for p in picks:
        cost += 0.12
        cost += box[p]
    assert cost <= 600.0
    # re-running with the picks as the seeded buffer and no budget left adds exactly the
    # unconditional initial pick (the reference appends it before any budget test)
    rc2, more = hip.greedy(D, picks.tolist(), -1, box, 0.12, cost, 600.0)
    assert rc2 == 0 and len(more) == 1 and more[0] not in set(picks.tolist())


# ---------------------------------------------------------------- uncertainty epilogues
def test_argsort_minmax_scale_entropy_primitives(oracle):
    from al3d import selector_ops as ops
    rng = np.random.default_rng(0)
    x = rng.normal(size=5000).astype(np.float32)
    x[[5, 77, 78, 4000]] = x[9]                      # ties
    x[[100, 2000]] = np.nan
    x[[7, 8]] = [np.inf, -np.inf]
    got = ops.argsort_desc(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
    assert np.array_equal(got, oracle.argsort_desc(x))
    assert np.array_equal(got, torch.argsort(-torch.from_numpy(x), stable=True).numpy())
    y = rng.uniform(0.1, 0.6, size=333).astype(np.float32)
    n = ops.minmax_norm(torch.from_numpy(y).to("cuda:0")).cpu().numpy()
    ty = torch.from_numpy(y)
    np.testing.assert_allclose(n, ((ty - ty.min()) / (ty.max() - ty.min())).numpy(), rtol=1e-6, atol=0)
    y[5] = np.nan
    assert np.isnan(ops.minmax_norm(torch.from_numpy(y).to("cuda:0")).cpu().numpy()).all()
    f = rng.normal(size=(40, 512)).astype(np.float32)
    w = rng.uniform(size=40).astype(np.float32)
    got = ops.scale_rows(torch.from_numpy(f).to("cuda:0"), torch.from_numpy(w).to("cuda:0")).cpu().numpy()
    assert np.array_equal(got, f * w[:, None])
    scores = rng.uniform(0.1, 0.99, size=(3, 6, 83)).astype(np.float32)
    counts = rng.integers(0, 84, size=(3, 6)).astype(np.int32)
    counts[1] = 0                                    # a frame without detections -> NaN
    e = ops.frame_entropy(torch.from_numpy(scores).to("cuda:0"), torch.from_numpy(counts).to("cuda:0")).cpu().numpy()
    for b in range(3):
        kept = np.concatenate([scores[b, t, :counts[b, t]] for t in range(6)])
        ref = oracle.frame_entropy(kept)
        if b == 1:
            assert np.isnan(e[b]) and np.isnan(ref)
        else:
            np.testing.assert_allclose(e[b], ref, rtol=2e-6)
            t = torch.from_numpy(kept)
            np.testing.assert_allclose(e[b], (-t * torch.log(t) - (1.0 - t) * torch.log(1 - t)).mean().item(), rtol=2e-6)


def test_ppal_primitives(oracle):
    from al3d import selector_ops as ops
    rng = np.random.default_rng(3)
    n = 300
    Dm = rng.uniform(0, 5, size=(n, n)).astype(np.float32)
    keep = rng.uniform(size=n) < 0.4
    got = ops.mask_map_(torch.from_numpy(Dm.copy()).to("cuda:0"), torch.from_numpy(keep.astype(np.uint8)).to("cuda:0")).cpu().numpy()
    ref = Dm.copy(); ref[~keep] = -np.inf; ref[:, ~keep] = -np.inf
    assert np.array_equal(got, ref)
    scores = rng.uniform(0.1, 0.99, size=(2, 6, 83)).astype(np.float32)
    labels = rng.integers(0, 10, size=(2, 6, 83)).astype(np.int32)
    counts = rng.integers(0, 84, size=(2, 6)).astype(np.int32)
    cw = rng.uniform(0.5, 2.0, size=10).astype(np.float32)
    e = ops.frame_weighted_entropy(*(torch.from_numpy(a).to("cuda:0") for a in (scores, labels, counts, cw))).cpu().numpy()
    for b in range(2):
        tot = 0.0
        for t in range(6):
            s = torch.from_numpy(scores[b, t, :counts[b, t]])
            h = -s * torch.log(s) - (1.0 - s) * torch.log(1 - s)
            tot += float((h * torch.from_numpy(cw[labels[b, t, :counts[b, t]]])).sum())
        np.testing.assert_allclose(e[b], tot, rtol=1e-5)


def test_l1_map_row_blocks_equal_full_map():
    """al3d_l1_distance_rows_f32: any row range of the map, bit for bit (the N>1 path's building block)."""
    import torch
    from al3d import lib
    g = torch.Generator(device="cpu").manual_seed(1)
    n, c = 333, 70                                     # not multiples of the tile or of 4
    feats = torch.randn(n, c, generator=g).cuda()
    full = torch.empty((n, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.call("al3d_l1_distance_f32", feats.data_ptr(), n, c, 2, full.data_ptr(), st)
    for r0, nr in [(0, 64), (64, 100), (200, 133), (332, 1), (0, n), (17, 0)]:
        blk = torch.full((max(nr, 1), n), float("nan"), dtype=torch.float32, device="cuda")
        lib.call("al3d_l1_distance_rows_f32", feats.data_ptr(), n, c, 2, r0, nr, blk.data_ptr(), st)
        if nr:
            assert torch.equal(blk[:nr].view(torch.int32), full[r0:r0 + nr].view(torch.int32)), (r0, nr)
    with pytest.raises(lib.Al3dError):
        lib.call("al3d_l1_distance_rows_f32", feats.data_ptr(), n, c, 2, 300, 100, full.data_ptr(), st)
