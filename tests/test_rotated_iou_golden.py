"""CPU suite: the rotated-box geometry behind the oracle's rotated NMS (row a10) against the reference's own second
implementation of it -- det3d/ops/nms/nms_gpu.py:183-420 (``rbbox_to_corners``, ``inter``, ``devRotateIoU``: numba.cuda device
functions, run as plain Python by oracle/gen_golden_rotated_iou.py) -- on 1,750 seeded pairs: corners to one or two float32
ulps (same formula; the Python run rounds a float64 cos / sin to float32 where C calls cosf / sinf), intersection area and IoU
to float32 rounding of two different algorithms (vertex collection + angular sort + triangle fan there, convex clip +
shoelace here).  The path the reference's test mode calls (nms_cpu.h, boost::geometry) is compiled code that cannot be built
here: its loop rule stays a restatement, its geometry is what this file cross-checks.

Where the two disagree the geometry is degenerate (coincident edges) and the reference's vertex collection is the unstable
one: on the same box twice it returns 0 for 66 of the 150 cases (its inside test fails on its own corners by rounding) and
overflows its 8-point buffer once; on one axis-aligned pair with a shared edge it returns 3.0 where the overlap is 2.0.
Those cases are checked against known answers instead (IoU 1; the exact overlap of two axis-aligned rectangles)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rotated_iou_pairs.npz")


def _axis_aligned_overlap(ca, cb):
    w = min(ca[:4].max(), cb[:4].max()) - max(ca[:4].min(), cb[:4].min())
    h = min(ca[4:].max(), cb[4:].max()) - max(ca[4:].min(), cb[4:].min())
    return max(w, 0.0) * max(h, 0.0)


def test_rotated_box_geometry_matches_the_references_python_implementation(oracle):
    z = np.load(GOLD)
    A, B = z["a"], z["b"]
    n = len(A)
    worst_area = worst_iou = 0.0
    compared = disagree = same = aligned = 0
    for i in range(n):
        ca, cb, inter, iou = oracle.rbox_pair(A[i], B[i])
        # the reference interleaves x0, y0, x1, y1, ...; the oracle keeps x0..x3 | y0..y3
        for mine, ref in ((ca, z["corners_a"][i]), (cb, z["corners_b"][i])):
            mine = np.stack([mine[:4], mine[4:]], 1).reshape(-1)
            assert np.all(np.abs(mine - ref) <= 4e-6 * np.maximum(1.0, np.abs(ref))), i
        scale = float(min(A[i][2] * A[i][3], B[i][2] * B[i][3]))
        right_angles = all(abs(v / (np.pi / 2) - round(v / (np.pi / 2))) < 1e-6 for v in (A[i][4], B[i][4]))
        if right_angles:                                  # known answer, independent of either implementation
            aligned += 1
            assert abs(inter - _axis_aligned_overlap(ca.astype(np.float64), cb.astype(np.float64))) <= 1e-5 * scale, i
        if np.array_equal(A[i], B[i]):                    # known answer: the same box twice
            same += 1
            assert abs(iou - 1.0) <= 1e-6, i
            continue
        if np.isnan(z["inter"][i]):
            continue                                      # the reference's 8-point vertex buffer overflowed on this pair
        compared += 1
        ea, ei = abs(inter - float(z["inter"][i])) / scale, abs(iou - float(z["iou"][i]))
        if ea > 2e-5 or ei > 2e-5:
            disagree += 1
            assert right_angles, (i, inter, float(z["inter"][i]))      # only where edges coincide (checked exactly above)
            continue
        worst_area, worst_iou = max(worst_area, ea), max(worst_iou, ei)
    print("pairs", n, "compared", compared, "same box", same, "axis-aligned", aligned, "disagree (degenerate)", disagree,
          "worst area error / smaller box", worst_area, "worst IoU error", worst_iou)
    assert compared >= 1590 and same == 150 and aligned >= 150 and disagree <= 2
