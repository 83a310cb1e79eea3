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

A second fixture pins what surrounds that call: ``MultiGroupHead.predict`` itself (mg_head.py:697-803) -- the view of each
task's head output as [anchors, 10] / [anchors, classes], ``GroundBox3dCoderTorch.decode_torch`` on ALL anchors
(box_coders.py:106-109), one ``get_task_detections`` per task and the merge of the six tasks with the label offsets -- called
unbound on seeded head outputs of a 16 x 16 map with the CBGS tasks' anchors (tests/golden/head_predict_tasks.npz).
Run in the build container only (needs /root/reference); writes tests/golden/head_predict.npz and head_predict_tasks.npz.
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
    predict_fixture(mg)


def predict_fixture(mg):
    """``MultiGroupHead.predict`` on the six CBGS tasks over a 16 x 16 map, two samples."""
    import functools
    ref_import.import_box_np_ops()
    coders = importlib.import_module("det3d.core.bbox.box_coders")
    sys.path.insert(0, os.path.dirname(HERE))
    from al3d.datasets.anchors import generate_task_anchors             # equal to the reference generator's: anchors golden
    from al3d.utils import Config
    cfgm = Config.fromfile(os.path.join(os.path.dirname(HERE), "examples", "active", "cbgs_spatial_temporal.py"))
    B, H, W = 2, 16, 16
    gens = [dict(g, anchor_ranges=[-12.8, -12.8, g["anchor_ranges"][2], 12.8, 12.8, g["anchor_ranges"][5]])
            for g in cfgm.target_assigner.anchor_generators]
    anchors = generate_task_anchors(cfgm.tasks, gens, [1, H, W])
    num_classes = [t["num_class"] for t in cfgm.tasks]
    na = [2 * c for c in num_classes]
    rng = np.random.default_rng(77)
    coder = coders.GroundBox3dCoderTorch(linear_dim=False, vec_encode=True, n_dim=9, norm_velo=False)
    head = types.SimpleNamespace(use_direction_classifier=False, encode_background_as_zeros=True, use_sigmoid_score=True,
                                 num_anchor_per_locs=na, num_classes=num_classes, anchor_dim=9, bev_only=False,
                                 box_n_dim=coder.code_size, box_coder=coder)
    head.get_task_detections = functools.partial(mg.MultiGroupHead.get_task_detections, head)
    cfg = _Cfg(score_threshold=0.1, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
               nms=_Cfg(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000, nms_post_max_size=83,
                        nms_iou_threshold=0.2))
    store = {"num_classes": np.array(num_classes, np.int64), "shape": np.array([B, H, W], np.int64)}
    preds = []
    for t, nc in enumerate(num_classes):
        box = rng.normal(0.0, 0.3, (B, H, W, na[t] * 10)).astype(np.float32)
        cls = rng.normal(-1.5 if t % 2 == 0 else -3.5, 2.0, (B, H, W, na[t] * nc)).astype(np.float32)
        if t == 3:
            cls[1] = -20.0                                                # one (sample, task) without any candidate
        preds.append({"box_preds": torch.from_numpy(box), "cls_preds": torch.from_numpy(cls)})
        store.update({f"box{t}": box, f"cls{t}": cls, f"anchors{t}": anchors[t]})
    example = {"voxels": None, "num_points": None, "coordinates": None,
               "anchors": [torch.from_numpy(np.broadcast_to(a, (B,) + a.shape).copy()) for a in anchors]}
    with torch.no_grad():
        out = mg.MultiGroupHead.predict(head, example, preds, cfg)
    for b in range(B):
        store.update({f"out{b}.boxes": out[b]["box3d_lidar"].numpy(), f"out{b}.scores": out[b]["scores"].numpy(),
                      f"out{b}.labels": out[b]["label_preds"].numpy()})
        print("sample", b, "->", tuple(out[b]["box3d_lidar"].shape), "labels", np.bincount(out[b]["label_preds"].numpy(), minlength=10))
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "head_predict_tasks.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
