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
#include <omp.h>

void al3d_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

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

/* ------------------------------------------------------------------ sparse conv
 * Rulebook + conv in the spconv formulation: per kernel offset k a list of (in,out) pairs
 * (getIndicePairsConv / getIndicePairsSubM, geometry.h:145-194,248-298) and
 * out[o] += in[i] @ W[k] for every pair, k ascending (indiceConv, spconv_ops.h:260-361).
 * Output sites of a strided conv are numbered in first-touch order like the CPU rulebook.
 * coords: [n,4] (b,z,y,x).  Returns the number of output rows. */
static int64_t cell_of(const int* c, const int* shape)
{
    return (((int64_t)c[0] * shape[0] + c[1]) * shape[1] + c[2]) * shape[2] + c[3];
}

int64_t al3d_oracle_spconv(const float* fin, const int* coords_in, int64_t n_in, int batch,
                           const int* in_shape, const float* wgt, int cin, int cout,
                           const int* ksize, const int* stride, const int* pad, int subm,
                           float* fout, int* coords_out, int64_t cap_out, int* out_shape)
{
    int k3[3] = {ksize[0], ksize[1], ksize[2]};
    int p3[3] = {pad[0], pad[1], pad[2]};
    if (subm) for (int d = 0; d < 3; ++d) p3[d] = k3[d] / 2;
    for (int d = 0; d < 3; ++d)
        out_shape[d] = subm ? in_shape[d] : (in_shape[d] + 2 * p3[d] - (k3[d] - 1) - 1) / stride[d] + 1;
    const int64_t ocells = (int64_t)batch * out_shape[0] * out_shape[1] * out_shape[2];
    int* grid = (int*)malloc(sizeof(int) * (size_t)ocells);
    memset(grid, 0xff, sizeof(int) * (size_t)ocells);
    int64_t n_out = 0;
    if (subm) {
        for (int64_t i = 0; i < n_in; ++i) {
            grid[cell_of(coords_in + 4 * i, out_shape)] = (int)i;
            memcpy(coords_out + 4 * i, coords_in + 4 * i, sizeof(int) * 4);
        }
        n_out = n_in;
    }
    const int K = k3[0] * k3[1] * k3[2];
    /* pairs per offset */
    int64_t* cnt = (int64_t*)calloc((size_t)K, sizeof(int64_t));
    int* pin = (int*)malloc(sizeof(int) * (size_t)(K * n_in + 1));
    int* pout = (int*)malloc(sizeof(int) * (size_t)(K * n_in + 1));
    for (int64_t i = 0; i < n_in; ++i) {
        const int* c = coords_in + 4 * i;
        for (int kz = 0; kz < k3[0]; ++kz) for (int ky = 0; ky < k3[1]; ++ky) for (int kx = 0; kx < k3[2]; ++kx) {
            const int kk[3] = {kz, ky, kx};
            int o[4] = {c[0], 0, 0, 0}, ok = 1;
            for (int d = 0; d < 3; ++d) {
                const int s = subm ? 1 : stride[d];
                const int t = c[1 + d] + p3[d] - kk[d];
                if (t < 0 || t % s) { ok = 0; break; }
                o[1 + d] = t / s;
                if (o[1 + d] >= out_shape[d]) { ok = 0; break; }
            }
            if (!ok) continue;
            const int64_t cell = cell_of(o, out_shape);
            if (grid[cell] == -1) {
                if (subm) continue;                 /* SubM: only existing sites are outputs */
                if (n_out >= cap_out) { free(grid); free(cnt); free(pin); free(pout); return -1; }
                memcpy(coords_out + 4 * n_out, o, sizeof(int) * 4);
                grid[cell] = (int)n_out++;
            }
            const int k = (kz * k3[1] + ky) * k3[2] + kx;
            pin[k * n_in + cnt[k]] = (int)i;
            pout[k * n_in + cnt[k]] = grid[cell];
            cnt[k]++;
        }
    }
    memset(fout, 0, sizeof(float) * (size_t)(n_out * cout));
    for (int k = 0; k < K; ++k) {
        const float* w = wgt + (int64_t)k * cin * cout;
        /* within one offset every output row appears at most once: pairs are independent */
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < cnt[k]; ++q) {
            const float* a = fin + (int64_t)pin[k * n_in + q] * cin;
            float* o = fout + (int64_t)pout[k * n_in + q] * cout;
            float acc[128];
            for (int co = 0; co < cout; ++co) acc[co] = 0.f;
            for (int ci = 0; ci < cin; ++ci) {             /* in_row @ W[k], ci-major (row-major W) */
                const float av = a[ci];
                const float* wr = w + (int64_t)ci * cout;
                for (int co = 0; co < cout; ++co) acc[co] += av * wr[co];
            }
            for (int co = 0; co < cout; ++co) o[co] += acc[co];
        }
    }
    free(grid); free(cnt); free(pin); free(pout);
    return n_out;
}

/* ------------------------------------------------------------------ box decode
 * second_box_decode, encode_angle_to_vector=True, smooth_dim=False, norm_velo=False
 * (det3d/core/bbox/box_torch_ops.py:80-148). enc [n,10], anchors [n,9] -> out [n,9]. */
void al3d_oracle_box_decode(const float* enc, const float* anc, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) {
        const float* t = enc + 10 * i; const float* a = anc + 9 * i; float* o = out + 9 * i;
        const float diag = sqrtf(a[4] * a[4] + a[3] * a[3]);
        o[0] = t[0] * diag + a[0];
        o[1] = t[1] * diag + a[1];
        o[2] = t[2] * a[5] + a[2];
        o[3] = expf(t[3]) * a[3];
        o[4] = expf(t[4]) * a[4];
        o[5] = expf(t[5]) * a[5];
        o[6] = t[6] + a[6];
        o[7] = t[7] + a[7];
        o[8] = atan2f(t[9] + sinf(a[8]), t[8] + cosf(a[8]));
    }
}

/* ------------------------------------------------------------------ rotated NMS
 * rotate_nms_cc + rotate_non_max_suppression_cpu (det3d/ops/nms/nms_cpu.py:34-45,
 * det3d/ops/nms/nms_cpu.h:73-168): corners (box_np_ops.py:479-499), standup boxes, skip pairs
 * with standup IoU <= 0, suppress when polygon IoU >= thresh.  boost::geometry is restated as
 * a convex clip (both polygons are rectangles) + shoelace; union = |A| + |B| - |A n B|.
 * dets [n,5] (x,y,w,l,r) ALREADY in descending score order.  Returns kept count. */
static void corners_of(const float* d, float* cx, float* cy)
{
    const float c = cosf(d[4]), s = sinf(d[4]);
    const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    for (int k = 0; k < 4; ++k) {
        const float px = ux[k] * d[2], py = uy[k] * d[3];
        cx[k] = px * c + py * s + d[0];
        cy[k] = -px * s + py * c + d[1];
    }
}

static float poly_area(const float* x, const float* y, int n)
{
    float a = 0.f;
    for (int k = 0; k < n; ++k) { int k2 = k + 1 == n ? 0 : k + 1; a += x[k] * y[k2] - x[k2] * y[k]; }
    return 0.5f * fabsf(a);
}

static float clip_area(const float* ax, const float* ay, const float* bx, const float* by)
{
    float px[16], py[16], qx[16], qy[16];
    int n = 4;
    for (int k = 0; k < 4; ++k) { px[k] = ax[k]; py[k] = ay[k]; }
    float barea = 0.f;
    for (int k = 0; k < 4; ++k) { int k2 = (k + 1) & 3; barea += bx[k] * by[k2] - bx[k2] * by[k]; }
    const float sgn = barea >= 0.f ? 1.f : -1.f;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const int e2 = (e + 1) & 3;
        const float ex = bx[e2] - bx[e], ey = by[e2] - by[e];
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const int k2 = k + 1 == n ? 0 : k + 1;
            const float d1 = sgn * (ex * (py[k] - by[e]) - ey * (px[k] - bx[e]));
            const float d2 = sgn * (ex * (py[k2] - by[e]) - ey * (px[k2] - bx[e]));
            if (d1 >= 0.f) { qx[m] = px[k]; qy[m] = py[k]; ++m; }
            if ((d1 >= 0.f) != (d2 >= 0.f)) {
                const float tt = d1 / (d1 - d2);
                qx[m] = px[k] + tt * (px[k2] - px[k]);
                qy[m] = py[k] + tt * (py[k2] - py[k]);
                ++m;
            }
        }
        n = m;
        for (int k = 0; k < n; ++k) { px[k] = qx[k]; py[k] = qy[k]; }
    }
    return n < 3 ? 0.f : poly_area(px, py, n);
}

/* The geometry of one pair, exported for the cross-check against the reference's OTHER implementation of the same quantity
 * (det3d/ops/nms/nms_gpu.py:343-420, `rbbox_to_corners` / `inter` / `devRotateIoU`, numba.cuda device functions run as plain
 * Python by oracle/gen_golden_rotated_iou.py): corners [x0..x3 | y0..y3] of both boxes, intersection area, IoU as the NMS
 * loop below forms it. */
void al3d_oracle_rbox_pair(const float* a5, const float* b5, float* corners_a, float* corners_b, float* inter, float* iou)
{
    corners_of(a5, corners_a, corners_a + 4);
    corners_of(b5, corners_b, corners_b + 4);
    const float in = clip_area(corners_a, corners_a + 4, corners_b, corners_b + 4);
    const float uni = poly_area(corners_a, corners_a + 4, 4) + poly_area(corners_b, corners_b + 4, 4) - in;
    *inter = in;
    *iou = uni > 0.f ? in / uni : 0.f;
}

int64_t al3d_oracle_rotate_nms(const float* dets, int64_t n, float thresh, int64_t post_max, int* keep)
{
    float* cx = (float*)malloc(sizeof(float) * 4 * (size_t)(n + 1));
    float* cy = (float*)malloc(sizeof(float) * 4 * (size_t)(n + 1));
    float* sb = (float*)malloc(sizeof(float) * 4 * (size_t)(n + 1));
    unsigned char* sup = (unsigned char*)calloc((size_t)(n + 1), 1);
    for (int64_t i = 0; i < n; ++i) {
        corners_of(dets + 5 * i, cx + 4 * i, cy + 4 * i);
        float x1 = cx[4 * i], x2 = x1, y1 = cy[4 * i], y2 = y1;
        for (int k = 1; k < 4; ++k) {
            x1 = fminf(x1, cx[4 * i + k]); x2 = fmaxf(x2, cx[4 * i + k]);
            y1 = fminf(y1, cy[4 * i + k]); y2 = fmaxf(y2, cy[4 * i + k]);
        }
        sb[4 * i] = x1; sb[4 * i + 1] = y1; sb[4 * i + 2] = x2; sb[4 * i + 3] = y2;
    }
    int64_t kept = 0;
    for (int64_t i = 0; i < n && kept < post_max; ++i) {
        if (sup[i]) continue;
        keep[kept++] = (int)i;
        const float ai = poly_area(cx + 4 * i, cy + 4 * i, 4);
        for (int64_t j = i + 1; j < n; ++j) {
            if (sup[j]) continue;
            const float iw = fminf(sb[4 * i + 2], sb[4 * j + 2]) - fmaxf(sb[4 * i], sb[4 * j]);
            const float ih = fminf(sb[4 * i + 3], sb[4 * j + 3]) - fmaxf(sb[4 * i + 1], sb[4 * j + 1]);
            if (!(iw > 0.f && ih > 0.f)) continue;
            const float inter = clip_area(cx + 4 * i, cy + 4 * i, cx + 4 * j, cy + 4 * j);
            if (!(inter > 0.f)) continue;
            const float uni = ai + poly_area(cx + 4 * j, cy + 4 * j, 4) - inter;
            if (uni > 0.f && inter / uni >= thresh) sup[j] = 1;
        }
    }
    free(cx); free(cy); free(sb); free(sup);
    return kept;
}

/* ---------------------------------------------------------------- a1: sweep merge
 * CPU restatement of read_file / remove_close / read_sweep / LoadPointCloudFromFile.__call__
 * (det3d/datasets/pipelines/loading.py:17-63,98-126); same argument meaning as
 * al3d_merge_sweeps_f32 (include/al3d.h).  Returns the number of rows written. */
int64_t al3d_oracle_merge_sweeps(const float* raw, const int64_t* file_off, int nfiles,
                                 const double* xform, const unsigned char* has_xform,
                                 const double* time_lag, float min_distance, float* out)
{
    int64_t n = 0;
    for (int f = 0; f < nfiles; ++f) {
        for (int64_t i = file_off[f]; i < file_off[f + 1]; ++i) {
            float x = raw[5 * i], y = raw[5 * i + 1], z = raw[5 * i + 2];
            if (f > 0 && fabsf(x) < min_distance && fabsf(y) < min_distance) continue;
            if (f > 0 && has_xform[f]) {
                const double* t = xform + 12 * f;
                const double xd = x, yd = y, zd = z;
                const float nx = (float)(((t[0] * xd + t[1] * yd) + t[2] * zd) + t[3]);
                const float ny = (float)(((t[4] * xd + t[5] * yd) + t[6] * zd) + t[7]);
                const float nz = (float)(((t[8] * xd + t[9] * yd) + t[10] * zd) + t[11]);
                x = nx; y = ny; z = nz;
            }
            float* o = out + 5 * n++;
            o[0] = x; o[1] = y; o[2] = z; o[3] = raw[5 * i + 3];
            o[4] = f == 0 ? 0.0f : (float)time_lag[f];
        }
    }
    return n;
}

/* ---------------------------------------------------------------- f4: BEV pooling (camera branch)
 * CPU restatement of BaseTransform.bev_pool (bevfusion/mmdet3d/models/vtransforms/base.py:127-163) around
 * bev_pool() / bev_pool_kernel (bevfusion/mmdet3d/ops/bev_pool/bev_pool.py:82-97, src/bev_pool_cuda.cu:21-44):
 * cell = trunc((geom - lo) / dx) per axis in float32 (lo = bx - dx/2), points outside [0, nx) dropped, the
 * C-vectors of a cell summed in float32.  The reference sums in the order its (unstable) argsort leaves the
 * points of a cell; this restatement -- like the device kernel -- uses ascending point index.
 * depth != NULL: the Lift-Splat outer product x[p] = depth[p] * ctx[pixel(p)] (depth_lss.py:92-97) with
 * x = ctx [BN,fH*fW,C], p = (bn*D + d)*fHW + pix; product rounded to float32 before the sum, as the reference's
 * materialised tensor.  out [B, nx0, nx1, nx2*C] must be zero-filled by the caller. */
void al3d_oracle_bev_pool(const float* x, const float* depth, int D, int fHW, const float* geom, int64_t P, int C,
                          int B, const float* lo, const float* dx, const int* nx, float* out)
{
    const int64_t per = P / B;
    for (int64_t p = 0; p < P; ++p) {
        int c3[3], ok = 1;
        for (int k = 0; k < 3; ++k) {
            const float t = (geom[3 * p + k] - lo[k]) / dx[k];
            if (!(t > -1.0f && t < (float)nx[k])) { ok = 0; break; }
            c3[k] = (int)t;
        }
        if (!ok) continue;
        const int64_t b = p / per;
        float* o = out + ((((b * nx[0] + c3[0]) * nx[1] + c3[1]) * nx[2]) + c3[2]) * (int64_t)C;
        if (depth) {
            const int64_t bn = p / ((int64_t)D * fHW), pix = p % fHW;
            const float* row = x + (bn * fHW + pix) * C;
            for (int c = 0; c < C; ++c) {
                const float v = depth[p] * row[c];
                o[c] += v;
            }
        } else {
            const float* row = x + p * C;
            for (int c = 0; c < C; ++c) o[c] += row[c];
        }
    }
}
