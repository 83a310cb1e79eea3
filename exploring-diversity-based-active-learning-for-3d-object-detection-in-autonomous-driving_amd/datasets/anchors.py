"""Anchor tables for the multi-group head -- host-side constants, generated once.

Semantics of the reference's ``create_anchors_3d_range`` (det3d/core/bbox/box_np_ops.py:735-807)
and ``TargetAssigner.generate_anchors`` (det3d/core/anchor/target_assigner.py:144-166): per
generator a ``[1,H,W,1,R,9]`` grid of (x, y, z, w, l, h, vx, vy, r); generators of one task are
concatenated along the per-location axis, giving anchor index ``(y*W + x)*na + cls*R + rot``.
The reference regenerates these for every sample (preprocess.py:346-378); here they are built
once and stay resident on the device.
"""
import numpy as np


def create_anchors_3d_range(feature_size, anchor_range, sizes, rotations, velocities=None,
                            dtype=np.float32):
    """-> [D, H, W, num_sizes, num_rots, 7|9] like the reference."""
    anchor_range = np.array(anchor_range, dtype)
    D, H, W = [int(s) for s in feature_size]
    stride = (anchor_range[3] - anchor_range[0]) / W
    z = np.linspace(anchor_range[2], anchor_range[5], D, dtype=dtype)
    y = np.linspace(anchor_range[1], anchor_range[4], H, endpoint=False, dtype=dtype) + stride / 2
    x = np.linspace(anchor_range[0], anchor_range[3], W, endpoint=False, dtype=dtype) + stride / 2
    rot = np.array(rotations, dtype=dtype)
    sizes = np.reshape(np.array(sizes, dtype=dtype), [-1, 3])
    extra = sizes
    if velocities is not None:
        vel = np.array(velocities, dtype=dtype).reshape([-1, 2])
        extra = np.hstack([sizes, vel]).reshape([-1, 5])
    ns, nr, ne = sizes.shape[0], rot.shape[0], extra.shape[1]
    out = np.empty((D, H, W, ns, nr, 4 + ne), dtype=dtype)
    out[..., 0] = x[None, None, :, None, None]
    out[..., 1] = y[None, :, None, None, None]
    out[..., 2] = z[:, None, None, None, None]
    out[..., 3:3 + ne] = extra[None, None, None, :, None, :]
    out[..., 3 + ne] = rot[None, None, None, None, :]
    return out


def generate_task_anchors(tasks, anchor_generators, feature_map_size):
    """Per task ``[H*W*na, 9]`` float32 (na = 2 * classes of the task).

    ``tasks``: list of dict(num_class, class_names); ``anchor_generators``: the config list of
    ``anchor_generator_range`` dicts (one per class, in class order)."""
    by_class = {g["class_name"]: g for g in anchor_generators}
    out = []
    for t in tasks:
        per_cls = []
        for name in t["class_names"]:
            g = by_class[name]
            a = create_anchors_3d_range(feature_map_size, g["anchor_ranges"], g["sizes"],
                                        g["rotations"], g.get("velocities"))
            per_cls.append(a.reshape([*a.shape[:3], -1, a.shape[-1]]))
        a = np.concatenate(per_cls, axis=-2)
        out.append(np.ascontiguousarray(a.reshape(-1, a.shape[-1])))
    return out
