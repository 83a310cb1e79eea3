/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the detector-side
 * arithmetic of the sweep: voxelisation + mean VFE, the spconv rulebook and
 * gather/GEMM/scatter, box decode and rotated NMS.  Never linked or called by the
 * product path (see oracle/al3d_oracle_selector.c for the rules).
 *
 * Parity status per function:
 *   voxelize        PINNED  (tests/golden/voxel_*.npz from the reference's own
 *                           points_to_voxel_new, oracle/gen_golden_detector.py)
 *   box decode      PINNED  (reference second_box_decode, same script)
 *   sparse conv     UNPINNED: spconv==1.2.1 is an external dependency absent from
 *                           /root/reference; restated from the in-tree vendored spconv-1.0
 *                           sources (bevfusion/mmdet3d/ops/spconv/include/spconv/geometry.h:25-82,
 *                           145-194,248-298; spconv_ops.h:260-361) and cross-checked against a
 *                           dense torch conv3d in tests/.
 *   rotated NMS     UNPINNED: det3d.ops.nms.nms needs boost::geometry (absent); restated
 *                           from det3d/ops/nms/nms_cpu.h:73-168 + nms_cpu.py:34-45 with
 *                           known-answer tests in tests/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* points_to_voxel_new + select_voxels (det3d/ops/point_cloud/point_cloud_ops.py:187-296)
 * + VoxelFeatureExtractorV3 (det3d/models/readers/voxel_encoder.py:206-211), one frame.
 * Sequential "first come" statement (the same semantics as the legacy C++ twin
 * det3d/ops/point_cloud/point_cloud_ops.h:12-71).  Returns the voxel count. */
int64_t al3d_oracle_voxelize(const float* pts, int64_t npts, int nfeat, const float* range_min,
                             const float* voxel_size, const int* grid, int max_points,
                             int max_voxels, float* voxels, int* coords_zyx, int* num_points,
                             float* feat)
{
    const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
    int* cell_to_voxel = (int*)malloc(sizeof(int) * (size_t)cells);
    memset(cell_to_voxel, 0xff, sizeof(int) * (size_t)cells);
    int64_t nvox = 0;
    for (int64_t i = 0; i < npts; ++i) {
        const float* p = pts + i * nfeat;
        int c[3], ok = 1;
        for (int d = 0; d < 3; ++d) {
            float f = floorf((p[d] - range_min[d]) / voxel_size[d]);
            if (!(f >= 0.f && f < (float)grid[d])) { ok = 0; break; }
            c[d] = (int)f;
        }
        if (!ok) continue;
        const int64_t cell = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
        int v = cell_to_voxel[cell];
        if (v == -1) {
            /* unique cells beyond max_voxels (by first appearance) are dropped with all
             * their points (point_cloud_ops.py:271-284) */
            if (nvox >= max_voxels) { cell_to_voxel[cell] = -2; continue; }
            v = (int)nvox++;
            cell_to_voxel[cell] = v;
            coords_zyx[3 * v + 0] = c[2]; coords_zyx[3 * v + 1] = c[1]; coords_zyx[3 * v + 2] = c[0];
            num_points[v] = 0;
            memset(voxels + (int64_t)v * max_points * nfeat, 0, sizeof(float) * (size_t)(max_points * nfeat));
        } else if (v == -2) continue;
        if (num_points[v] < max_points) {
            memcpy(voxels + ((int64_t)v * max_points + num_points[v]) * nfeat, p, sizeof(float) * (size_t)nfeat);
            num_points[v]++;
        }
    }
    for (int64_t v = 0; v < nvox; ++v)
        for (int f = 0; f < nfeat; ++f) {
            float s = 0.f;
            for (int t = 0; t < max_points; ++t) s += voxels[(v * max_points + t) * nfeat + f];
            feat[v * nfeat + f] = s / (float)num_points[v];
        }
    free(cell_to_voxel);
    return nvox;
}
