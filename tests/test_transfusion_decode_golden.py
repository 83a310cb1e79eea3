"""CPU suite: ``TransFusionHead._decode`` (host torch on a few hundred proposals) against the reference's own
``TransFusionBBoxCoder.decode(..., filter=True)`` (bevfusion/mmdet3d/core/bbox/coders/transfusion_bbox_coder.py:39-123), run by
oracle/gen_golden_bevfusion_models.py on seeded inputs: boxes, scores and labels of every surviving proposal, bit for bit."""
import os
import types

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bevfusion_transfusion_decode.npz")


def test_transfusion_decode_equals_the_reference_coder():
    from al3d.models.transfusion_head import TransFusionHead
    z = np.load(GOLD)
    t = lambda k: torch.from_numpy(z[k])                                 # noqa: E731
    coder = dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                 post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=float(z["score_threshold"][0]))
    got = TransFusionHead._decode(types.SimpleNamespace(bbox_coder=coder), t("heat"), t("rot"), t("dim"), t("center"),
                                  t("height"), t("vel"))
    assert len(got) == 2
    kept = 0
    for i, r in enumerate(got):
        assert torch.equal(r["bboxes"], t(f"out{i}.bboxes")) and torch.equal(r["scores"], t(f"out{i}.scores"))
        assert torch.equal(r["labels"], t(f"out{i}.labels"))
        kept += len(r["scores"])
    assert 0 < kept < 2 * z["heat"].shape[2]                              # the filters removed something, not everything
