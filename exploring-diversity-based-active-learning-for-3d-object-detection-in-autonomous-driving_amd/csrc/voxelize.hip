// Point-cloud voxelisation + mean VFE for a batch of frames, deterministic.
//
// Reference semantics (det3d/ops/point_cloud/point_cloud_ops.py:213-296 +
// det3d/models/readers/voxel_encoder.py:206-211):
//   c = floor((p_xyz - range_min) / voxel_size) in float32; keep 0 <= c < grid
//   voxels are numbered in order of FIRST APPEARANCE in the point list; only the first
//   max_voxels of them are kept; each keeps its first max_points points in input order;
//   feature = sum(points of the voxel, slot order) / count.
// A parallel device has no "first come": the order is rebuilt from point indices --
//   (1) atomicMin of the point index per cell finds each cell's first point ("leader"),
//   (2) an exclusive scan over leader flags numbers the voxels in first-appearance order,
//   (3) points are bucketed per voxel, and one thread per voxel picks its max_points
//       smallest point indices in ascending order.
// Every output is therefore independent of scheduling.  Built with -ffp-contract=off:
// the float32 subtract/divide/floor must round like numpy.
#include "al3d_common.h"
#include "al3d_scan.h"

#define VX_EMPTY 0x7fffffff

struct VoxCfg {
    float min_x, min_y, min_z, vs_x, vs_y, vs_z;
    int gx, gy, gz;         // grid size (x, y, z)
    int max_points, max_voxels, nfeat;
};

__device__ __forceinline__ int64_t vox_cell(const float* __restrict__ p, const VoxCfg& c)
{
    const float fx = floorf((p[0] - c.min_x) / c.vs_x);
    const float fy = floorf((p[1] - c.min_y) / c.vs_y);
    const float fz = floorf((p[2] - c.min_z) / c.vs_z);
    if (!(fx >= 0.f && fx < (float)c.gx && fy >= 0.f && fy < (float)c.gy && fz >= 0.f && fz < (float)c.gz))
        return -1;
    return ((int64_t)(int)fz * c.gy + (int)fy) * c.gx + (int)fx;
}

__device__ __forceinline__ int frame_of(const int64_t* __restrict__ off, int B, int64_t i)
{
    int lo = 0, hi = B;  // largest b with off[b] <= i
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// pass 1: cell id per point; first[b][cell] = min point index (frame-local)
__global__ void vox_first_kernel(const float* __restrict__ pts, const int64_t* __restrict__ off, int B,
                                 VoxCfg c, int64_t cells, int* __restrict__ first, int* __restrict__ pcell_hi,
                                 int* __restrict__ pcell_lo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t npts = off[B];
    // frame of the wave's FIRST point by a wave-uniform search (scalar loads), then a step forward for the lanes beyond a
    // frame boundary: seven dependent vector loads per point otherwise
    const int64_t iw = i - (threadIdx.x & 63);
    const int64_t i0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(iw >> 32)) << 32) |
                       (unsigned)__builtin_amdgcn_readfirstlane((int)iw);
    if (i0 >= npts) return;
    int b = frame_of(off, B, i0);
    if (i >= npts) return;
    while (b + 1 < B && off[b + 1] <= i) ++b;
    // x, y, z with one 12-byte load (points are nfeat floats apart: dword-aligned)
    typedef float vx_f32x3 __attribute__((ext_vector_type(3)));
    vx_f32x3 xyz;
    asm volatile("global_load_dwordx3 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(xyz) : "v"(pts + i * c.nfeat) : "memory");
    const float p3[3] = {xyz[0], xyz[1], xyz[2]};
    const int64_t cell = vox_cell(p3, c);
    // cells < 2^31 is checked on the host, so one int carries the cell id
    const int ci = cell < 0 ? -1 : (int)cell;
    pcell_lo[i] = ci;
    pcell_hi[i] = b;
    // A ring scan puts neighbouring returns into the same voxel: a lane whose (frame, cell) also sits in one of the four
    // lanes below it has the larger point index, so its atomicMin could not change the cell's minimum -- it is skipped
    // (the lowest lane of such a group always issues).  Lanes past the end of the list have left; their shuffles return the
    // reader's own value, which the lane test discards.
    const int lane = threadIdx.x & 63;
    bool dup = false;
#pragma unroll
    for (int k = 1; k <= 4; ++k) {
        const int c2 = __shfl_up(ci, k), b2 = __shfl_up(b, k);
        dup = dup || (lane >= k && c2 == ci && b2 == b);
    }
    if (ci >= 0 && !dup) atomicMin(&first[(int64_t)b * cells + ci], (int)(i - off[b]));
}

// pass 2: leader flags
// also records every point's leader (plead: pass 4 then needs no second random probe of the 22 GB grid)
__global__ void vox_flag_kernel(const int64_t* __restrict__ off, int B, int64_t cells,
                                const int* __restrict__ first, const int* __restrict__ pframe,
                                const int* __restrict__ pcell, int* __restrict__ flag, int* __restrict__ plead)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= off[B]) return;
    const int cell = pcell[i], b = pframe[i];
    // runs of adjacent lanes in one (frame, cell): the run's first lane probes the grid, the others take its answer
    const int lane = threadIdx.x & 63;
    const int c1 = __shfl_up(cell, 1), b1 = __shfl_up(b, 1);
    const bool same = lane > 0 && cell >= 0 && c1 == cell && b1 == b;
    const unsigned long long starts = __ballot(!same);
    const int start = 63 - __builtin_clzll(starts & ((2ull << lane) - 1ull));
    int lead = (cell >= 0 && !same) ? first[(int64_t)b * cells + cell] : -1;
    lead = __shfl(lead, start);
    plead[i] = lead;
    flag[i] = (cell >= 0 && lead == (int)(i - off[b])) ? 1 : 0;
}

// pass 3: voxels per frame, row bases (frames are concatenated in the outputs)
__global__ void vox_base_kernel(const int64_t* __restrict__ off, int B, const int* __restrict__ lscan,
                                const int* __restrict__ flag, int max_voxels, int* __restrict__ num_voxels,
                                int* __restrict__ row_base)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t s = off[b], e = off[b + 1];
        int tot = 0;
        if (e > s) tot = lscan[e - 1] + flag[e - 1] - lscan[s];
        if (tot > max_voxels) tot = max_voxels;
        num_voxels[b] = tot;
        row_base[b] = base;
        base += tot;
    }
    row_base[B] = base;
}

// pass 4: output row per point (or -1), coordinates of kept voxels, points per voxel
// prow holds the point's leader on entry (vox_flag_kernel) and its output row on exit (same thread, same word)
__global__ void vox_assign_kernel(const int64_t* __restrict__ off, int B, VoxCfg c, int64_t cells,
                                  const int* __restrict__ pframe,
                                  const int* __restrict__ pcell, const int* __restrict__ lscan,
                                  const int* __restrict__ row_base, int* __restrict__ prow,
                                  int* __restrict__ ppos, int* __restrict__ coords, int* __restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= off[B]) return;
    const int cell = pcell[i], b = pframe[i];
    int row = -1, pos = 0;
    const int lead = cell >= 0 ? prow[i] : -1;
    // runs of adjacent lanes of one voxel (same frame and leader): the run's first lane looks the row up and takes the
    // run's slots with ONE atomicAdd, the others derive theirs (slot order inside a voxel is irrelevant: pass 6 sorts)
    const int lane = threadIdx.x & 63;
    const int l1 = __shfl_up(lead, 1), b1 = __shfl_up(b, 1);
    const bool same = lane > 0 && cell >= 0 && l1 == lead && b1 == b;
    const unsigned long long act = __ballot(true), starts = __ballot(!same);
    const unsigned long long upto = (2ull << lane) - 1ull;
    const int start = 63 - __builtin_clzll(starts & upto);
    const unsigned long long later = starts & ~upto;
    const int end = later ? (int)__builtin_ctzll(later) : 64 - (int)__builtin_clzll(act);
    if (cell >= 0 && !same) {
        const int vid = lscan[off[b] + lead] - lscan[off[b]];
        if (vid < c.max_voxels) {
            row = row_base[b] + vid;
            pos = atomicAdd(&cnt[row], end - start);   // arrival positions inside the voxel: the bucket slots (pass 5)
        }
    }
    row = __shfl(row, start);
    pos = __shfl(pos, start) + (lane - start);
    if (row < 0) pos = 0;
    if (cell >= 0) {
        if (row >= 0) {
            if (lead == (int)(i - off[b])) {
                const int x = cell % c.gx, y = (cell / c.gx) % c.gy, z = cell / (c.gx * c.gy);
                *reinterpret_cast<int4*>(coords + 4 * (int64_t)row) = make_int4(b, z, y, x);
            }
        }
    }
    prow[i] = row;
    ppos[i] = pos;
}

// pass 5: bucket the point indices per voxel at the arrival positions pass 4's counter handed out (arrival
// order is irrelevant: pass 6 sorts) -- no second round of atomics
__global__ void vox_bucket_kernel(const int64_t* __restrict__ off, int B, const int* __restrict__ prow,
                                  const int* __restrict__ ppos, const int* __restrict__ boff, int* __restrict__ bucket)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the point list ends at off[B]: `points` may hold more rows than that (a loader's compaction leaves an unused tail),
    // and passes 1-4 never wrote the per-point arrays beyond it
    if (i >= off[B]) return;
    const int row = prow[i];
    if (row < 0) return;
    bucket[boff[row] + ppos[i]] = (int)i;
}

// pass 6: one thread per voxel: its max_points smallest point indices, ascending;
// padded voxel tensor, clipped count and the mean feature.
#define VX_MAXP 32
// One thread per voxel: keep the MAXP smallest point indices of the voxel's bucket in a sorted
// register array (compile-time indexed compare-exchange chain -- a runtime-indexed array would live
// in scratch memory), then read each kept point's features once and accumulate the per-feature sums
// in ascending point order (the order the reference's reduction sees).
template <int MAXP, int NFEAT>
__global__ __launch_bounds__(128) void vox_gather_fixed_kernel(const float* __restrict__ pts, int rows,
                                  const int* __restrict__ boff, const int* __restrict__ cnt,
                                  const int* __restrict__ bucket, const int* __restrict__ row_base,
                                  int B, float* __restrict__ voxels,
                                  int* __restrict__ num_points, float* __restrict__ feat)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows || row >= row_base[B]) return;
    const int n = cnt[row], o = boff[row];
    const int keep = n < MAXP ? n : MAXP;
    int sel[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) sel[i] = 0x7fffffff;
    for (int t = 0; t < n; ++t) {
        int idx = bucket[o + t];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {       // sorted insert: carry the larger value down the chain
            const int lo = idx < sel[i] ? idx : sel[i];
            const int hi = idx < sel[i] ? sel[i] : idx;
            sel[i] = lo;
            idx = hi;
        }
    }
    num_points[row] = keep;
    float sum[NFEAT];
#pragma unroll
    for (int f = 0; f < NFEAT; ++f) sum[f] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXP; ++t) {
        float v[NFEAT];
#pragma unroll
        for (int f = 0; f < NFEAT; ++f) v[f] = 0.f;
        if (t < keep) {
            const float* pp = pts + (int64_t)sel[t] * NFEAT;
#pragma unroll
            for (int f = 0; f < NFEAT; ++f) v[f] = pp[f];
        }
#pragma unroll
        for (int f = 0; f < NFEAT; ++f) sum[f] += v[f];
        if (voxels) {
#pragma unroll
            for (int f = 0; f < NFEAT; ++f) voxels[((int64_t)row * MAXP + t) * NFEAT + f] = v[f];
        }
    }
    const float denom = (float)keep;
#pragma unroll
    for (int f = 0; f < NFEAT; ++f) feat[(int64_t)row * NFEAT + f] = sum[f] / denom;
}

__global__ void vox_gather_kernel(const float* __restrict__ pts, VoxCfg c, int rows,
                                  const int* __restrict__ boff, const int* __restrict__ cnt,
                                  const int* __restrict__ bucket, const int* __restrict__ row_base,
                                  int B, float* __restrict__ voxels,
                                  int* __restrict__ num_points, float* __restrict__ feat)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows || row >= row_base[B]) return;
    const int n = cnt[row], o = boff[row];
    const int keep = n < c.max_points ? n : c.max_points;
    int sel[VX_MAXP];
    int have = 0;
    for (int t = 0; t < n; ++t) {           // insertion into a sorted list of <= max_points
        const int idx = bucket[o + t];
        if (have == c.max_points && idx > sel[have - 1]) continue;
        int pos = have < c.max_points ? have : c.max_points - 1;
        while (pos > 0 && sel[pos - 1] > idx) { sel[pos] = sel[pos - 1]; --pos; }
        sel[pos] = idx;
        if (have < c.max_points) ++have;
    }
    num_points[row] = keep;
    const float denom = (float)keep;
    for (int f = 0; f < c.nfeat; ++f) {
        float s = 0.f;
        for (int t = 0; t < c.max_points; ++t) {
            const float v = t < keep ? pts[(int64_t)sel[t] * c.nfeat + f] : 0.f;
            if (voxels) voxels[((int64_t)row * c.max_points + t) * c.nfeat + f] = v;
            s += v;
        }
        feat[(int64_t)row * c.nfeat + f] = s / denom;
    }
}

// pass 7: put the touched cells of the first-index grid back to EMPTY
// one write per occupied cell: its leader restores it (every point of the cell used to)
__global__ void vox_restore_kernel(const int64_t* __restrict__ off, int B, int64_t cells, const int* __restrict__ pframe,
                                   const int* __restrict__ pcell, const int* __restrict__ flag, int* __restrict__ first)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= off[B]) return;                                 // (as vox_bucket_kernel: rows past off[B] were never classified)
    if (flag[i]) first[(int64_t)pframe[i] * cells + pcell[i]] = VX_EMPTY;
}

__global__ void vox_fill_i32(int* p, int64_t n, int v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

extern "C" int64_t al3d_voxelize_grid_bytes(int B, int gx, int gy, int gz)
{
    return (int64_t)B * gx * gy * gz * 4;
}

extern "C" int al3d_voxelize_grid_init(void* grid, int B, int gx, int gy, int gz, void* stream)
{
    AL3D_REQUIRE(grid, "al3d_voxelize_grid_init: null grid");
    const int64_t n = (int64_t)B * gx * gy * gz;
    hipLaunchKernelGGL(vox_fill_i32, dim3(2048), dim3(256), 0, (hipStream_t)stream, (int*)grid, n, VX_EMPTY);
    AL3D_CHECK_LAUNCH("vox_fill_i32");
    return AL3D_OK;
}

extern "C" int64_t al3d_voxelize_workspace_bytes(int64_t npts, int B, int max_voxels)
{
    const int64_t rows = (int64_t)B * max_voxels;
    // pframe, pcell, flag, lscan, prow, bucket, ppos (npts each) + cnt, boff (rows+1 each)
    // + scan scratch
    return al3d_align(npts * 4, 256) * 7 + al3d_align((rows + 1) * 4, 256) * 2 +
           al3d_scan_workspace_bytes(npts > rows ? npts : rows) + 1024;
}

extern "C" int al3d_voxelize_mean_f32(const float* points, const int64_t* point_offsets, int64_t npts,
                                      int B, int nfeat, const float* range_min, const float* voxel_size,
                                      const int* grid_size, int max_points, int max_voxels,
                                      void* first_grid, void* workspace, float* feat, int* coords,
                                      int* num_points, float* voxels, int* num_voxels, int* row_base,
                                      void* stream)
{
    AL3D_REQUIRE(points && point_offsets && range_min && voxel_size && grid_size && first_grid &&
                     workspace && feat && coords && num_points && num_voxels && row_base,
                 "al3d_voxelize_mean_f32: null pointer");
    AL3D_REQUIRE(B >= 1 && nfeat >= 3 && npts >= 0 && npts < (1LL << 31), "al3d_voxelize_mean_f32: bad sizes");
    AL3D_REQUIRE(max_points >= 1 && max_points <= VX_MAXP, "al3d_voxelize_mean_f32: max_points must be in [1,%d]", VX_MAXP);
    AL3D_REQUIRE(max_voxels >= 1, "al3d_voxelize_mean_f32: max_voxels must be >= 1");
    AL3D_REQUIRE(((uintptr_t)coords & 15) == 0 && ((uintptr_t)points & 3) == 0,
                 "al3d_voxelize_mean_f32: coords must be 16-byte aligned (one int4 per voxel), points 4-byte aligned");
    VoxCfg c;
    c.min_x = range_min[0]; c.min_y = range_min[1]; c.min_z = range_min[2];
    c.vs_x = voxel_size[0]; c.vs_y = voxel_size[1]; c.vs_z = voxel_size[2];
    c.gx = grid_size[0]; c.gy = grid_size[1]; c.gz = grid_size[2];
    c.max_points = max_points; c.max_voxels = max_voxels; c.nfeat = nfeat;
    const int64_t cells = (int64_t)c.gx * c.gy * c.gz;
    AL3D_REQUIRE(cells > 0 && cells < (1LL << 31), "al3d_voxelize_mean_f32: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * max_voxels;
    unsigned char* w = (unsigned char*)workspace;
    auto take = [&](int64_t bytes) { unsigned char* p = w; w += al3d_align(bytes, 256); return p; };
    int* pframe = (int*)take(npts * 4);
    int* pcell = (int*)take(npts * 4);
    int* flag = (int*)take(npts * 4);
    int* lscan = (int*)take(npts * 4);
    int* prow = (int*)take(npts * 4);
    int* bucket = (int*)take(npts * 4);
    int* cnt = (int*)take((rows + 1) * 4);
    int* boff = (int*)take((rows + 1) * 4);
    int* ppos = (int*)take(npts * 4);
    void* scan_ws = (void*)w;
    int* first = (int*)first_grid;
    const unsigned pb = (unsigned)al3d_cdiv(npts > 0 ? npts : 1, 256);
    if (hipMemsetAsync(cnt, 0, (rows + 1) * 4, s) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "al3d_voxelize_mean_f32: memset failed");
    hipLaunchKernelGGL(vox_first_kernel, dim3(pb), dim3(256), 0, s, points, point_offsets, B, c, cells,
                       first, pframe, pcell);
    hipLaunchKernelGGL(vox_flag_kernel, dim3(pb), dim3(256), 0, s, point_offsets, B, cells, first, pframe,
                       pcell, flag, prow);
    int rc = al3d_exclusive_scan_i32(flag, lscan, npts, scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_base_kernel, dim3(1), dim3(64), 0, s, point_offsets, B, lscan, flag, max_voxels,
                       num_voxels, row_base);
    hipLaunchKernelGGL(vox_assign_kernel, dim3(pb), dim3(256), 0, s, point_offsets, B, c, cells,
                       pframe, pcell, lscan, row_base, prow, ppos, coords, cnt);
    rc = al3d_exclusive_scan_i32(cnt, boff, rows + 1, scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_bucket_kernel, dim3(pb), dim3(256), 0, s, point_offsets, B, prow, ppos, boff, bucket);
    if (c.max_points == 10 && c.nfeat == 5)      // the nuScenes configuration: register-resident selection
        hipLaunchKernelGGL((vox_gather_fixed_kernel<10, 5>), dim3((unsigned)al3d_cdiv(rows, 128)), dim3(128), 0, s,
                           points, (int)rows, boff, cnt, bucket, row_base, B, voxels, num_points, feat);
    else
        hipLaunchKernelGGL(vox_gather_kernel, dim3((unsigned)al3d_cdiv(rows, 128)), dim3(128), 0, s, points, c,
                           (int)rows, boff, cnt, bucket, row_base, B, voxels, num_points, feat);
    hipLaunchKernelGGL(vox_restore_kernel, dim3(pb), dim3(256), 0, s, point_offsets, B, cells, pframe, pcell, flag, first);
    AL3D_CHECK_LAUNCH("voxelize");
    return AL3D_OK;
}

// VoxelFeatureExtractorV3 on reference-format input (padded voxels [M,max_points,F] + counts):
// features[:, :, :F].sum(1) / num_points  (det3d/models/readers/voxel_encoder.py:206-211).
__global__ void vfe_mean_kernel(const float* __restrict__ voxels, const int* __restrict__ num, int m,
                                int max_points, int nfeat, float* __restrict__ feat)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)m * nfeat) return;
    const int row = (int)(e / nfeat), f = (int)(e % nfeat);
    float s = 0.f;
    for (int t = 0; t < max_points; ++t) s += voxels[((int64_t)row * max_points + t) * nfeat + f];
    feat[e] = s / (float)num[row];
}

extern "C" int al3d_vfe_mean_f32(const float* voxels, const int* num_points, int m, int max_points,
                                 int nfeat, float* feat, void* stream)
{
    AL3D_REQUIRE(voxels && num_points && feat && m >= 0, "al3d_vfe_mean_f32: bad arguments");
    if (m == 0) return AL3D_OK;
    hipLaunchKernelGGL(vfe_mean_kernel, dim3((unsigned)al3d_cdiv((int64_t)m * nfeat, 256)), dim3(256), 0,
                       (hipStream_t)stream, voxels, num_points, m, max_points, nfeat, feat);
    AL3D_CHECK_LAUNCH("vfe_mean_kernel");
    return AL3D_OK;
}
