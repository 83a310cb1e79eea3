"""CPU suite: the oracle (plain-C restatement) against the golden vectors that the
reference's own selector classes produced (oracle/gen_golden_selectors.py)."""
import glob
import hashlib
import os

import numpy as np
import pytest

import selector_logic as L
from al3d import synthetic

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "selector_*.npz")))


def _feats(fx):
    if "feats_seed" not in fx:
        return None
    f = synthetic.make_embeddings(fx["n_boxes"].shape[0], seed=int(fx["feats_seed"]),
                                  scale=float(fx["feats_scale"]))
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(fx["feats_sha256"]), \
        "synthetic.make_embeddings drifted from the generator the fixtures were made with"
    return f


def test_fixture_inventory():
    assert len(CASES) >= 19


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[9:-4] for p in CASES])
def test_oracle_matches_reference(oracle, path):
    fx = L.load_case(path)
    maps = {}
    rc, full = L.run_case(fx, oracle, feats=_feats(fx), maps_out=maps)
    if str(fx["error"]) == "AssertionError":
        assert rc == -1, "reference raised its duplicate-pick assertion here"
        return
    assert rc == 0
    # integer work: the selected-index buffer must equal the reference's, in order
    assert full == fx["selected"].tolist()
    # float64 maps: bit-exact where the reference's numpy exp and ours agree (they do on
    # every fixture); documented bound is 1 ulp (numpy's exp is CPU-dependent).
    if "ref_raw_spatial_map" in fx and "raw_spatial" in maps:
        assert L.ulp_diff_f64(maps["raw_spatial"], fx["ref_raw_spatial_map"]).max() == 0
    if "ref_distance_map" in fx and "distance_map" in maps:
        ref = fx["ref_distance_map"]
        if ref.dtype == np.float32:
            # float32 L1 map: torch's vectorised reduction order differs from ours
            np.testing.assert_allclose(maps["distance_map"], ref, rtol=2e-6, atol=0)
        elif "feats_seed" in fx:
            # f64 sum containing the f32 feature term: tolerance of that term
            np.testing.assert_allclose(maps["distance_map"], ref, rtol=0, atol=4e-6)
        else:
            assert L.ulp_diff_f64(maps["distance_map"], ref).max() <= 1


def test_exp_against_libm(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.uniform(0, 60, 200000), rng.uniform(-745, 709, 20000),
                        [0.0, -0.0, -1e6, -745.2, 709.78, 1e-300, -1e-300]])
    got = oracle.exp_f64(x)
    ref = np.exp(x)
    ok = np.isfinite(ref) & (ref > 1e-300)
    assert L.ulp_diff_f64(got[ok], ref[ok]).max() <= 1
    assert got[np.where(x == -1e6)[0][0]] == 0.0
    assert np.isnan(oracle.exp_f64(np.array([np.nan])))[0]
    assert oracle.exp_f64(np.array([np.inf]))[0] == np.inf
    assert oracle.exp_f64(np.array([-np.inf]))[0] == 0.0


def test_apsp_equals_scipy(oracle):
    """Dijkstra restatement vs scipy on a random kNN graph (same construction as the
    reference, spatial_temporal_selector.py:92-104) -- must agree bit for bit."""
    from scipy import sparse, spatial
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 100, size=(300, 2))
    xy[7] = xy[3]  # coincident positions: zero-length edge == no edge
    tree = spatial.cKDTree(xy)
    kd, ki = tree.query(xy, 9)
    W = np.zeros((300, 300))
    for a, (d, i) in enumerate(zip(kd, ki)):
        W[a, i] = d
        W[i, a] = d
    ref = sparse.csgraph.shortest_path(W, directed=False, method="D")
    got = oracle.spatial_map(xy, 8)
    assert L.ulp_diff_f64(got, ref).max() == 0


def test_greedy_tie_rule_and_budget(oracle):
    # all-equal map: every argmax tie must resolve to the lowest index
    n = 16
    D = np.ones((n, n))
    np.fill_diagonal(D, 0.0)
    box = np.zeros(n)
    rc, picks = oracle.greedy(D, [], 5, box, 0.12, 0.0, 1.0)
    # cost 0.12 per frame, budget 1 -> 8 frames fit (0.96), the 9th overflows
    assert rc == 0 and picks.tolist() == [5, 0, 1, 2, 3, 4, 6, 7]
    # exhausting the pool trips the duplicate assertion like the reference (A.1 #13)
    rc, picks = oracle.greedy(D, [], 0, box, 0.12, 0.0, 100.0)
    assert rc == -1 and len(picks) == n


def test_combine_rows_equals_rows_of_the_full_map(oracle):
    """The row-block form the large-pool checkers use (bench.verify_selection, test_config2_*) gives
    the bits of the full-map form, for every normalisation / aggregation."""
    rng = np.random.default_rng(8)
    n = 97
    S = rng.uniform(0, 60, (n, n))
    S[rng.random((n, n)) < 0.1] = np.inf
    F = rng.uniform(0, 3, (n, n)).astype(np.float32)
    run_id = np.sort(rng.integers(0, 5, n)).astype(np.int64)
    for norm in ("exp", "linear", "none"):
        for agg in ("sum", "min", "max"):
            kw = dict(normalize=norm, aggregate=agg, lambda_t=0.7, lambda_f=1.3, spatial_scale=55.0,
                      temporal_scale=31.0)
            full = oracle.combine(n, spatial=S, temporal_id=run_id, feat=F, **kw)
            blk = oracle.combine_rows(n, 40, spatial_rows=S[40:53], temporal_id=run_id, feat_rows=F[40:53], **kw)
            assert np.array_equal(full[40:53].view(np.int64), blk.view(np.int64)), (norm, agg)
    st = oracle.combine_rows(n, 5, spatial_rows=S[5:6], temporal_id=run_id)
    assert np.array_equal(st.view(np.int64), oracle.combine(n, spatial=S, temporal_id=run_id)[5:6].view(np.int64))
