"""CPU suite: the oracle's restatement of the head's post-processing (row a9: scores, best class, ``>=`` threshold, top-k into
the NMS, result indexing, range filter, label / score bookkeeping) against the reference's own
``MultiGroupHead.get_task_detections`` (det3d/models/bbox_heads/mg_head.py:805-1080), run by oracle/gen_golden_head_predict.py
on seeded (decoded boxes, class logits).  The compiled polygon NMS inside that call is this build's oracle in BOTH (see the
generator's docstring), so what is pinned is the reference's Python around it: same boxes in the same order (bit for bit --
they are gathered inputs), same labels, scores to one float32 ulp (torch.sigmoid against a float64 evaluation)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_predict.npz")
RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]


@pytest.mark.parametrize("case", ["two_class", "one_class", "nothing_passes"])
def test_task_detections_equal_the_reference_head(oracle, case):
    z = np.load(GOLD)
    thr, pre, post, iou = z[f"{case}.cfg"]
    b, s, l = oracle.task_detections(z[f"{case}.boxes"], z[f"{case}.logits"], float(thr), float(iou), int(pre), int(post), RANGE)
    rb, rs, rl = z[f"{case}.out_boxes"], z[f"{case}.out_scores"], z[f"{case}.out_labels"]
    assert b.shape == rb.shape and np.array_equal(b, rb), (b.shape, rb.shape)
    assert np.array_equal(l, rl)
    assert np.all(np.abs(s - rs) <= 1.2e-7 * np.maximum(np.abs(rs), 1e-3))
    if case == "nothing_passes":
        assert len(b) == 0
    else:
        assert 10 < len(b) <= post and np.all(np.diff(s) <= 0)               # survivors best first
        assert np.all(np.abs(b[:, :2]) <= 61.2)                              # the range filter removed the far ones


def test_predict_over_six_tasks_equals_the_reference_head(oracle):
    """``MultiGroupHead.predict`` of the reference (mg_head.py:697-803) on seeded head outputs of a 16 x 16 map: the oracle's
    per-task view of the head output + decode of the candidates + NMS glue, merged with the label offsets, gives the same
    detections in the same order.  Boxes: C float32 decode against torch's (2e-6, angle 1e-5, as in the decode golden)."""
    z = np.load(os.path.join(os.path.dirname(GOLD), "head_predict_tasks.npz"))
    B, H, W = (int(v) for v in z["shape"])
    ncs = [int(v) for v in z["num_classes"]]
    off = np.concatenate([[0], np.cumsum(ncs)])
    for b in range(B):
        bb, ss, ll = [], [], []
        for t, nc in enumerate(ncs):
            na = 2 * nc
            hout = np.concatenate([z[f"box{t}"][b].reshape(H * W, -1), z[f"cls{t}"][b].reshape(H * W, -1)], axis=1)
            bx, sc, lb = oracle.head_predict(hout, z[f"anchors{t}"], na, nc, 0, na * 10, 0.1, 0.2, 1000, 83, RANGE)
            bb.append(bx); ss.append(sc); ll.append(lb + off[t])
        bb, ss, ll = np.concatenate(bb), np.concatenate(ss), np.concatenate(ll)
        rb, rs, rl = z[f"out{b}.boxes"], z[f"out{b}.scores"], z[f"out{b}.labels"]
        assert ll.tolist() == rl.tolist() and len(rl) > 300
        assert np.all(np.abs(ss - rs) <= 1.2e-7 * np.maximum(np.abs(rs), 1e-3))
        np.testing.assert_allclose(bb[:, :8], rb[:, :8], rtol=2e-6, atol=2e-6)
        d = np.abs(bb[:, 8] - rb[:, 8])
        assert np.minimum(d, 2 * np.pi - d).max() < 1e-5
    assert not np.any(z["out1.labels"] == off[3])                                # the (sample, task) without a candidate
