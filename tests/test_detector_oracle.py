"""CPU suite: detector-side oracle pieces against the reference's golden vectors (decode,
anchors), against an independent dense torch conv3d (sparse conv -- spconv itself is absent,
parity unpinned), and known-answer tests for the rotated NMS restatement."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

G = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def random_sparse(rng, batch, shape, n, c):
    cells = set()
    while len(cells) < n:
        cells.add((int(rng.integers(batch)), int(rng.integers(shape[0])), int(rng.integers(shape[1])),
                   int(rng.integers(shape[2]))))
    coords = np.array(sorted(cells), dtype=np.int32)
    rng.shuffle(coords)
    feats = rng.normal(size=(len(coords), c)).astype(np.float32)
    return feats, coords


def to_dense(feats, coords, batch, shape):
    d = np.zeros((batch, feats.shape[1], *shape), dtype=np.float32)
    d[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]] = feats
    return d


@pytest.mark.parametrize("subm,k,s,p", [(True, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
                                         (False, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                         (False, (3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                         (False, (3, 1, 1), (2, 1, 1), (0, 0, 0)),
                                         (False, (1, 1, 3), (1, 1, 2), (0, 0, 0)),
                                         (True, (3, 3, 1), (1, 1, 1), (0, 0, 0)),
                                         (False, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                         (False, (2, 2, 2), (2, 2, 2), (0, 0, 0))])
def test_oracle_spconv_equals_dense_conv3d(oracle, subm, k, s, p):
    rng = np.random.default_rng(11)
    shape, batch, cin, cout = [11, 14, 12], 2, 5, 7
    feats, coords = random_sparse(rng, batch, shape, 300, cin)
    w = rng.normal(size=(*k, cin, cout)).astype(np.float32)
    fout, cout_c, oshape = oracle.spconv(feats, coords, batch, shape, w, k, s, p, subm)
    dense_in = torch.from_numpy(to_dense(feats, coords, batch, shape))
    wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2).contiguous()
    if subm:
        ref = F.conv3d(dense_in, wt, padding=[q // 2 for q in k]).numpy()
        assert np.array_equal(cout_c, coords)           # SubM keeps sites and their order
    else:
        ref = F.conv3d(dense_in, wt, stride=s, padding=p).numpy()
        mask = F.max_pool3d(torch.from_numpy(to_dense(np.ones((len(coords), 1), np.float32), coords,
                                                      batch, shape)), k, s, p).numpy()[:, 0] > 0
        got_mask = np.zeros_like(mask)
        got_mask[cout_c[:, 0], cout_c[:, 1], cout_c[:, 2], cout_c[:, 3]] = True
        assert np.array_equal(mask, got_mask)           # output sites = dilated input sites
    assert list(ref.shape[2:]) == oshape
    got = ref[cout_c[:, 0], :, cout_c[:, 1], cout_c[:, 2], cout_c[:, 3]]
    np.testing.assert_allclose(fout, got, rtol=1e-4, atol=1e-4)


def test_oracle_decode_matches_reference(oracle):
    z = np.load(os.path.join(G, "decode.npz"))
    got = oracle.box_decode(z["enc"].reshape(-1, 10), z["anchors"].reshape(-1, 9))
    ref = z["dec"].reshape(-1, 9)
    np.testing.assert_allclose(got[:, :8], ref[:, :8], rtol=2e-6, atol=2e-6)
    d = np.abs(got[:, 8] - ref[:, 8])
    assert np.minimum(d, 2 * np.pi - d).max() < 1e-5    # angles compare modulo 2 pi


def test_anchor_tables_match_reference():
    from al3d.datasets.anchors import generate_task_anchors
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal.py"))
    z = np.load(os.path.join(G, "anchors.npz"))
    tabs = generate_task_anchors(cfg.tasks, cfg.target_assigner.anchor_generators, [1, 128, 128])
    assert len(tabs) == 6
    for t, a in enumerate(tabs):
        assert a.shape == (int(z[f"task{t}_count"]), 9) and a.dtype == np.float32
        assert sha(a) == str(z[f"task{t}_sha256"])       # constants: bit-exact
        assert np.array_equal(a[:64], z[f"task{t}_head"]) and np.array_equal(a[-64:], z[f"task{t}_tail"])


def test_nms_known_answers(oracle):
    # identical boxes: the second is suppressed (IoU 1 >= 0.2)
    d = np.array([[0, 0, 2, 4, 0.3], [0, 0, 2, 4, 0.3]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0]
    # far apart: both kept (standup boxes do not overlap -> pair skipped)
    d = np.array([[0, 0, 2, 4, 0.0], [10, 0, 2, 4, 0.0]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0, 1]
    # touching edges: intersection has zero area -> not suppressed
    d = np.array([[0, 0, 2, 2, 0.0], [2, 0, 2, 2, 0.0]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0, 1]
    # axis-aligned half overlap: IoU = 1/3 >= 0.2 suppressed, but kept at thresh 0.5
    d = np.array([[0, 0, 2, 2, 0.0], [1, 0, 2, 2, 0.0]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0]
    assert oracle.rotate_nms(d, 0.5, 83).tolist() == [0, 1]
    # unit square vs the same square rotated 45 deg: intersection is a regular octagon,
    # IoU = (2*sqrt(2)-2) / (2 - (2*sqrt(2)-2)) = 0.7071
    d = np.array([[0, 0, 1, 1, 0.0], [0, 0, 1, 1, np.pi / 4]], np.float32)
    assert oracle.rotate_nms(d, 0.70, 83).tolist() == [0]
    assert oracle.rotate_nms(d, 0.72, 83).tolist() == [0, 1]
    # contained box: IoU = area ratio 1/16 < 0.2 -> kept
    d = np.array([[0, 0, 4, 4, 0.2], [0.1, 0.1, 1, 1, 0.2]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0, 1]
    # degenerate zero-area box never suppresses nor gets suppressed
    d = np.array([[0, 0, 0, 0, 0.0], [0, 0, 2, 2, 0.0], [0, 0, 2, 2, 0.1]], np.float32)
    assert oracle.rotate_nms(d, 0.2, 83).tolist() == [0, 1]
    # post_max truncation keeps the first survivors
    d = np.array([[10 * i, 0, 2, 2, 0.0] for i in range(10)], np.float32)
    assert oracle.rotate_nms(d, 0.2, 3).tolist() == [0, 1, 2]


def test_corner_convention_matches_reference():
    z = np.load(os.path.join(G, "anchors.npz"))
    d, ref = z["corner_in"], z["corner_out"]
    for row, want in zip(d, ref):
        c, s = np.cos(row[4]), np.sin(row[4])
        ux = np.array([-0.5, -0.5, 0.5, 0.5]) * row[2]
        uy = np.array([-0.5, 0.5, 0.5, -0.5]) * row[3]
        got = np.stack([ux * c + uy * s + row[0], -ux * s + uy * c + row[1]], 1)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_model_state_dict_layout_matches_reference():
    """Parameter names / shapes the reference checkpoints use."""
    from al3d.models import build_detector
    from al3d.utils import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(G), "..", "examples", "active",
                                       "cbgs_spatial_temporal.py"))
    m = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = m.state_dict()
    assert tuple(sd["backbone.middle_conv0.0.weight"].shape) == (3, 3, 3, 5, 16)
    assert tuple(sd["backbone.middle_conv0.3.conv1.weight"].shape) == (3, 3, 3, 16, 16)
    assert "backbone.middle_conv0.3.conv1.bias" in sd and "backbone.middle_conv0.0.bias" not in sd
    assert tuple(sd["backbone.middle_conv3.2.weight"].shape) == (3, 1, 1, 128, 128)
    assert tuple(sd["neck.blocks.0.1.weight"].shape) == (128, 256, 3, 3)
    assert tuple(sd["neck.deblocks.1.0.weight"].shape) == (256, 256, 2, 2)
    assert tuple(sd["bbox_head.tasks.1.conv_box.weight"].shape) == (40, 512, 1, 1)
    assert tuple(sd["bbox_head.tasks.5.conv_cls.bias"].shape) == (8,)
    z = np.load(os.path.join(G, "rpn.npz"))
    neck_keys = sorted(k[len("neck."):] for k in sd if k.startswith("neck."))
    assert neck_keys == sorted(str(k) for k in z["state_keys"])
