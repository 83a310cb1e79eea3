"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the post-processing around the rotated NMS (row a9): the reference's own
``MultiGroupHead.get_task_detections`` (det3d/models/bbox_heads/mg_head.py:805-1080, class-agnostic branch as the CBGS configs
run it: sigmoid scores, best class per anchor, ``>=`` score threshold, ``box_torch_ops.rotate_nms`` = top-k by score then the
rotated NMS, post-centre-range filter) called as an unbound method on seeded (decoded boxes, class logits) of one task.

One piece of that call chain is compiled code that does not exist here: ``rotate_nms_cc`` (det3d/ops/nms/nms_cpu.py:34-45 ->
nms_cpu.h, boost).  It is replaced by a function that sorts by score as ``rotate_nms_cc`` does and asks THIS BUILD'S oracle
(al3d_oracle_rotate_nms) which boxes survive -- so the fixture pins the reference's Python around the NMS (what is selected
into it, in which order, how its result is indexed back, the range filter, label / score bookkeeping), not the polygon clipping
(cross-checked separately: oracle/gen_golden_rotated_iou.py).  Other stand-ins are content-free: empty modules for the model
zoo's packages, placeholder names for the loss / checkpoint helpers mg_head.py imports, an attribute dict for the config.
Run in the build container only (needs /root/reference); writes tests/golden/head_predict.npz.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
import ref_import  # noqa: E402

ROOT = ref_import.REFERENCE_ROOT


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _rotate_nms_cc(dets, thresh):
    """Stand-in for the compiled call: same contract as nms_cpu.py:34-45 (dets [n, 6] = x, y, w, l, r, score -> kept indices
    into ``dets``, best first), the polygon part by this build's oracle."""
    order = dets[:, 5].argsort()[::-1].astype(np.int64)
    kept = oracle.rotate_nms(dets[order][:, :5], float(thresh), len(dets))
    return order[kept]


def import_head():
    ref_import.install_standins()
    bto = ref_import.import_box_torch_ops()
    sys.modules["det3d.core"].box_torch_ops = bto             # `from det3d.core import box_torch_ops` on the package shell
    ref_import._mod("det3d.ops.nms.nms_cpu", rotate_nms_cc=_rotate_nms_cc)
    import det3d.torchie  # noqa: F401
    for pkg in ("det3d.models", "det3d.models.bbox_heads", "det3d.models.losses"):
        if pkg not in sys.modules:
            ref_import._pkg(pkg, os.path.join(ROOT, *pkg.split(".")))
    ref_import._mod("det3d.models.builder", build_loss=None)
    sys.modules["det3d.models"].builder = sys.modules["det3d.models.builder"]
    ref_import._mod("det3d.models.losses.metrics")
    sys.modules["det3d.models.losses"].metrics = sys.modules["det3d.models.losses.metrics"]
    sys.modules["det3d.models.losses"].accuracy = None
    ref_import._mod("det3d.models.registry", HEADS=types.SimpleNamespace(register_module=lambda cls: cls))
    if "det3d.torchie.cnn" not in sys.modules:
        ref_import._mod("det3d.torchie.cnn", constant_init=None, kaiming_init=None)
    if "det3d.torchie.trainer" not in sys.modules:
        ref_import._mod("det3d.torchie.trainer", load_checkpoint=None)
    return importlib.import_module("det3d.models.bbox_heads.mg_head"), bto


def main():
    oracle.build()
    mg, _ = import_head()
    rng = np.random.default_rng(31)
    store = {}
    cases = {"two_class": dict(n=6000, nc=2, thr=0.1, pre=1000, post=83, iou=0.2),
             "one_class": dict(n=3000, nc=1, thr=0.3, pre=200, post=50, iou=0.2),
             "nothing_passes": dict(n=500, nc=2, thr=0.999999, pre=1000, post=83, iou=0.2)}
    for name, c in cases.items():
        n, nc = c["n"], c["nc"]
        boxes = np.zeros((n, 9), np.float32)
        boxes[:, :2] = rng.uniform(-70, 70, (n, 2))                  # some centres beyond the +-61.2 m range
        boxes[: n // 3, :2] = rng.normal(0, 6, (n // 3, 2))          # a dense cluster: many suppressions
        boxes[:, 2] = rng.uniform(-6, 4, n)
        boxes[:, 3:6] = np.exp(rng.normal(0.6, 0.4, (n, 3)))
        boxes[:, 6:8] = rng.normal(0, 1, (n, 2))
        boxes[:, 8] = rng.uniform(-np.pi, np.pi, n)
        logits = rng.normal(-3.0, 2.0, (n, nc)).astype(np.float32)
        cfg = _Cfg(score_threshold=c["thr"], post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                   nms=_Cfg(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=c["pre"],
                            nms_post_max_size=c["post"], nms_iou_threshold=c["iou"]))
        head = types.SimpleNamespace(use_direction_classifier=False, encode_background_as_zeros=True, use_sigmoid_score=True,
                                     num_anchor_per_locs=[2], num_classes=[nc], anchor_dim=9)
        with torch.no_grad():
            out = mg.MultiGroupHead.get_task_detections(head, 0, nc, cfg, torch.from_numpy(logits)[None],
                                                        torch.from_numpy(boxes)[None], [None], [None], [None])[0]
        store.update({f"{name}.boxes": boxes, f"{name}.logits": logits,
                      f"{name}.cfg": np.array([c["thr"], c["pre"], c["post"], c["iou"]], np.float64),
                      f"{name}.out_boxes": out["box3d_lidar"].numpy(), f"{name}.out_scores": out["scores"].numpy(),
                      f"{name}.out_labels": out["label_preds"].numpy()})
        print(name, "->", tuple(out["box3d_lidar"].shape))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "head_predict.npz")
    np.savez_compressed(out, **store)
    print("wrote", out)


if __name__ == "__main__":
    main()
