// BEV pooling of the camera branch (reference
// bevfusion/mmdet3d/models/vtransforms/base.py:127-163 + bevfusion/mmdet3d/ops/bev_pool/bev_pool.py:82-97 +
// src/bev_pool_cuda.cu:21-44): every frustum point p carries a C-vector x[p] and a lidar-frame position;
// points that fall into the same BEV cell are summed, out[b, ix, iy, iz*C + c] (the reference's
// [B,C,D,H,W] -> cat(unbind(2), 1) layout, channels last).
//
// The reference sorts the points by cell rank (torch.argsort, unstable) and sums each run with one thread per
// (run, channel).  Here nothing is sorted globally: cell ids -> per-cell counts (integer atomics: order-free) ->
// exclusive scan -> per-cell member lists filled in arrival order -> one wave per cell puts its (short) list into
// ascending point order by rank counting and sums in that order.  The summation order is therefore FIXED
// (ascending p, i.e. what a stable sort would give) and the result deterministic; the reference's own order is
// implementation-defined, so parity with it is a floating-point tolerance either way.
//
// Two forms: x materialised ([P,C], the reference's exact op) and the Lift-Splat outer product fused in
// (x[p] = depth[p] * ctx[pixel(p)], depth_lss.py:92-97) so that the [B,N,D,fH,fW,C] tensor -- 638 MB per sample at
// 6 x 118 x 32 x 88 x 80 -- is never written: the pooling then reads 13 MB.
#include "al3d_common.h"
#include "al3d_scan.h"

struct BevGrid {
    float lo[3], dx[3];       // lo = bx - dx/2 (float32, as the reference computes it), cell size
    int nx[3];                // cells along x, y, z
    int B;
    int64_t per_batch;        // points per sample
};

// ((g - lo) / dx).long(): truncation toward zero; kept iff 0 <= cell < nx (base.py:136,147-154)
__device__ __forceinline__ int bev_axis_cell(float g, float lo, float dx, int nx)
{
    const float t = (g - lo) / dx;
    if (!(t > -1.0f && t < (float)nx)) return -1;          // also drops NaN
    return (int)t;
}

__global__ void bev_cell_kernel(const float* __restrict__ geom, int64_t P, BevGrid g, int* __restrict__ cell,
                                int* __restrict__ count)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int ix = bev_axis_cell(geom[3 * p], g.lo[0], g.dx[0], g.nx[0]);
    const int iy = bev_axis_cell(geom[3 * p + 1], g.lo[1], g.dx[1], g.nx[1]);
    const int iz = bev_axis_cell(geom[3 * p + 2], g.lo[2], g.dx[2], g.nx[2]);
    int c = -1;
    if (ix >= 0 && iy >= 0 && iz >= 0) {
        const int b = (int)(p / g.per_batch);
        c = ((b * g.nx[0] + ix) * g.nx[1] + iy) * g.nx[2] + iz;
        atomicAdd(&count[c], 1);
    }
    cell[p] = c;
}

__global__ void bev_scatter_kernel(const int* __restrict__ cell, int64_t P, const int* __restrict__ start,
                                   int* __restrict__ fill, int* __restrict__ list)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int c = cell[p];
    if (c < 0) return;
    list[start[c] + atomicAdd(&fill[c], 1)] = (int)p;
}

#define BEV_SHORT 128
#define BEV_LONG_CAP 8192

// Long member lists: one 256-thread workgroup per cell with more than BEV_SHORT points sorts its list (bitonic, in LDS)
// into `sorted`.  The cells are found by a grid-stride scan over the counts (no host round trip); lists beyond
// BEV_LONG_CAP entries fall back to rank counting by the whole workgroup.
__global__ __launch_bounds__(256) void bev_sort_long_kernel(const int* __restrict__ count, const int* __restrict__ start,
                                                            const int* __restrict__ list, int* __restrict__ sorted,
                                                            int ncell)
{
    __shared__ int key[BEV_LONG_CAP];
    for (int c = blockIdx.x; c < ncell; c += gridDim.x) {
        const int len = count[c], base = start[c];
        if (len <= BEV_SHORT) continue;                                  // workgroup-uniform
        if (len <= BEV_LONG_CAP) {
            int np2 = 1;
            while (np2 < len) np2 <<= 1;
            for (int i = threadIdx.x; i < np2; i += 256) key[i] = i < len ? list[base + i] : 0x7fffffff;
            __syncthreads();
            for (int size = 2; size <= np2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int i = threadIdx.x; i < np2; i += 256) {
                        const int j = i ^ stride;
                        if (j > i) {
                            const bool asc = (i & size) == 0;
                            const int a = key[i], b = key[j];
                            if (asc ? a > b : a < b) { key[i] = b; key[j] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = threadIdx.x; i < len; i += 256)
                __hip_atomic_store(&sorted[base + i], key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        } else {
            for (int e = threadIdx.x; e < len; e += 256) {
                const int v = list[base + e];
                int rank = 0;
                for (int j = 0; j < len; ++j) rank += list[base + j] < v ? 1 : 0;
                __hip_atomic_store(&sorted[base + rank], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// One wave per cell.  LSS = 1: x[p][c] = depth[p] * ctx[row(p)][c] with p = ((bn*D + d)*fH + h)*fW + w and
// row = (bn*fH + h)*fW + w; product and sum are separate fp32 roundings (no contraction), like the reference's
// materialised tensor.
template <int LSS>
__global__ __launch_bounds__(256) void bev_sum_kernel(const float* __restrict__ x, const float* __restrict__ depth,
                                                      int C, int Dd, int fHW, const int* __restrict__ count,
                                                      const int* __restrict__ start, const int* __restrict__ list,
                                                      int* __restrict__ sorted, int ncell, int nz,
                                                      float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c0 >= ncell) return;
    const int len = count[c0], base = start[c0];
    // ascending point order.  Short lists (the common case: ~14 points per cell) by rank counting inside the wave;
    // long ones (cells next to a camera hold ~1,000 points) were put in order by bev_sort_long_kernel already
    // (rank counting is O(len^2 / 64) per wave: 16 k trips at len = 1,000).
    if (len <= BEV_SHORT) {
        for (int e = lane; e < len; e += 64) {
            const int v = list[base + e];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += list[base + j] < v ? 1 : 0;
            // L2-scope store / loads below: neighbouring cells share cache lines of this array and a CU's vector L1
            // is not refreshed by stores (MI355X_MICROARCH.md, inter-workgroup visibility)
            __hip_atomic_store(&sorted[base + rank], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // out[(b, ix, iy), iz*C + c]: cell id = ((b*nx0 + ix)*nx1 + iy)*nz + iz, so the address is cell * C + c
    float* o = out + (int64_t)c0 * C;
    // lanes = channels (two rounds at C = 80 .. 128); the member indices are fetched 64 at a time (one L2-scope load
    // per lane) and broadcast, the feature loads of four consecutive members are issued together and added in order
    const int nround = (C + 63) / 64;
    for (int rd = 0; rd < nround; ++rd) {
        const int c = rd * 64 + lane;
        const bool live = c < C;
        float acc = 0.f;
        for (int i0 = 0; i0 < len; i0 += 64) {
            const int mine = i0 + lane < len ? __hip_atomic_load(&sorted[base + i0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            const int cnt = len - i0 < 64 ? len - i0 : 64;
            for (int k0 = 0; k0 < cnt; k0 += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u < cnt ? k0 + u : cnt - 1;
                    const int p = __shfl(mine, k);
                    float t = 0.f;
                    if (live) {
                        if (LSS) {
                            const int bn = p / (Dd * fHW), pix = p % fHW;
                            t = depth[p] * x[((int64_t)bn * fHW + pix) * C + c];
                        } else {
                            t = x[(int64_t)p * C + c];
                        }
                    }
                    v[u] = k0 + u < cnt ? t : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (k0 + u < cnt) acc += v[u];
            }
        }
        if (live) o[c] = acc;
    }
    (void)nz;
}

// The same sums with lanes = float4 channel groups and SEVERAL cells per wave (C % 4 == 0: the shipped shapes, C = 80).
// G = C / 4 lanes own one cell (three cells per wave at C = 80), so one wave-instruction adds one member to up to
// four cells instead of 64 lanes adding one member to one cell in ceil(C / 64) rounds; the member list of a cell is
// put in ascending point order in LDS (rank counting inside the lane group; long lists arrive sorted from
// bev_sort_long_kernel, 128 at a time) together with what the inner loop needs per member -- the context row and
// the depth value -- computed ONCE per member by one lane (the generic kernel above divides p by D*fH*fW and fH*fW
// in every lane for every member: it is bound by those integer divisions).  Same products, same order of additions
// per (cell, channel): bit-identical to bev_sum_kernel.
#define BEV_CPW 4                                                // cells per wave at most
template <int LSS>
__global__ __launch_bounds__(256) void bev_sum_vec_kernel(const float* __restrict__ x, const float* __restrict__ depth,
                                                          int C, int G, int CPW, int Dd, int fHW,
                                                          const int* __restrict__ count, const int* __restrict__ start,
                                                          const int* __restrict__ list, const int* __restrict__ sorted,
                                                          int ncell, float* __restrict__ out)
{
    __shared__ int s_key[4][BEV_CPW][BEV_SHORT];
    __shared__ int s_row[4][BEV_CPW][BEV_SHORT];
    __shared__ float s_dep[4][BEV_CPW][BEV_SHORT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane / G, sub = lane - grp * G;
    const int64_t cell = ((int64_t)blockIdx.x * 4 + w) * CPW + grp;
    const bool own = grp < CPW && cell < ncell;
    const int len = own ? count[cell] : 0, base = own ? start[cell] : 0;
    int maxlen = 0;
    for (int g = 0; g < CPW; ++g) { const int l = __shfl(len, g * G); maxlen = l > maxlen ? l : maxlen; }
    const int gi = grp < CPW ? grp : 0;
    int* key = s_key[w][gi];
    int* rowv = s_row[w][gi];
    float* depv = s_dep[w][gi];
    const int ddfhw = Dd * fHW;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < maxlen; c0 += BEV_SHORT) {
        int clen = len - c0;
        clen = clen < 0 ? 0 : (clen > BEV_SHORT ? BEV_SHORT : clen);
        if (len <= BEV_SHORT) {
            for (int e = sub; e < clen; e += G) key[e] = list[base + e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int e = sub; e < clen; e += G) {
                const int v = key[e];
                int rank = 0;
                for (int j = 0; j < clen; ++j) rank += key[j] < v ? 1 : 0;
                if (LSS) {
                    const int bn = v / ddfhw, pix = v % fHW;
                    rowv[rank] = bn * fHW + pix;
                    depv[rank] = depth[v];
                } else {
                    rowv[rank] = v;
                }
            }
        } else {
            for (int e = sub; e < clen; e += G) {
                const int v = sorted[base + c0 + e];
                if (LSS) {
                    const int bn = v / ddfhw, pix = v % fHW;
                    rowv[e] = bn * fHW + pix;
                    depv[e] = depth[v];
                } else {
                    rowv[e] = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int cmax = 0;
        for (int g = 0; g < CPW; ++g) { const int l = __shfl(clen, g * G); cmax = l > cmax ? l : cmax; }
        for (int i = 0; i < cmax; i += 4) {
            float4 t[4];
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                d[u] = 0.f;
                if (i + u < clen) {
                    t[u] = *reinterpret_cast<const float4*>(x + (int64_t)rowv[i + u] * C + sub * 4);
                    if (LSS) d[u] = depv[i + u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < clen) {
                    if (LSS) {
                        acc.x += d[u] * t[u].x; acc.y += d[u] * t[u].y; acc.z += d[u] * t[u].z; acc.w += d[u] * t[u].w;
                    } else {
                        acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w;
                    }
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (own) *reinterpret_cast<float4*>(out + cell * C + sub * 4) = acc;
}

extern "C" int64_t al3d_bev_pool_workspace_bytes(int64_t n_points, int64_t n_cells)
{
    const int64_t p = n_points > 0 ? n_points : 1, c = n_cells > 0 ? n_cells : 1;
    return 3 * al3d_align(p * 4, 256) + 3 * al3d_align((c + 1) * 4, 256) + al3d_scan_workspace_bytes(c + 1);
}

struct BevPlan { int *cell, *list, *sorted, *count, *start, *fill; void* scan_ws; int64_t cb; };

static BevPlan bev_plan_carve(void* workspace, int64_t P, int ncell)
{
    unsigned char* w = (unsigned char*)workspace;
    const int64_t pb = al3d_align(P * 4, 256), cb = al3d_align(((int64_t)ncell + 1) * 4, 256);
    BevPlan p;
    p.cell = (int*)w;
    p.list = (int*)(w + pb);
    p.sorted = (int*)(w + 2 * pb);
    p.count = (int*)(w + 3 * pb);
    p.start = (int*)(w + 3 * pb + cb);
    p.fill = (int*)(w + 3 * pb + 2 * cb);
    p.scan_ws = w + 3 * pb + 3 * cb;
    p.cb = cb;
    return p;
}

static int bev_check_grid(int64_t P, int C, int B, const float* lo, const float* dx, const int* nx, const char* name, int* ncell)
{
    AL3D_REQUIRE(P >= 0 && P < (1LL << 31) && C >= 1 && B >= 1 && nx, "%s: bad arguments", name);
    AL3D_REQUIRE(nx[0] >= 1 && nx[1] >= 1 && nx[2] >= 1 && P % B == 0, "%s: bad grid / points not divisible by B", name);
    const int64_t ncell64 = (int64_t)B * nx[0] * nx[1] * nx[2];
    AL3D_REQUIRE(ncell64 < (1LL << 31) && ncell64 * C < (1LL << 40), "%s: grid too large", name);
    *ncell = (int)ncell64;
    return AL3D_OK;
}

// the part that depends on the geometry only: cell of every point, members of every cell in ascending point order
static int bev_pool_plan(const float* geom, int64_t P, int B, const float* lo, const float* dx, const int* nx, void* workspace,
                         hipStream_t s, const char* name)
{
    int ncell;
    int rc = bev_check_grid(P, 1, B, lo, dx, nx, name, &ncell);
    if (rc) return rc;
    AL3D_REQUIRE(geom && lo && dx && workspace && P > 0, "%s: null pointer / no point", name);
    const BevPlan p = bev_plan_carve(workspace, P, ncell);
    if (hipMemsetAsync(p.count, 0, (size_t)p.cb, s) != hipSuccess || hipMemsetAsync(p.fill, 0, (size_t)p.cb, s) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "%s: memset failed", name);
    BevGrid g;
    for (int k = 0; k < 3; ++k) { g.lo[k] = lo[k]; g.dx[k] = dx[k]; g.nx[k] = nx[k]; }
    g.B = B; g.per_batch = P / B;
    const unsigned pblocks = (unsigned)al3d_cdiv(P, 256);
    hipLaunchKernelGGL(bev_cell_kernel, dim3(pblocks), dim3(256), 0, s, geom, P, g, p.cell, p.count);
    rc = al3d_exclusive_scan_i32(p.count, p.start, (int64_t)ncell + 1, p.scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(bev_scatter_kernel, dim3(pblocks), dim3(256), 0, s, p.cell, P, p.start, p.fill, p.list);
    hipLaunchKernelGGL(bev_sort_long_kernel, dim3(1024), dim3(256), 0, s, p.count, p.start, p.list, p.sorted, ncell);
    AL3D_CHECK_LAUNCH(name);
    return AL3D_OK;
}

// the part that depends on the features: per cell the sum of its members' rows in ascending point order
static int bev_pool_apply(const float* x, const float* depth, int lss, int Dd, int fHW, int64_t P, int C, int B, const int* nx,
                          float* out, void* workspace, hipStream_t s, const char* name)
{
    int ncell;
    int rc = bev_check_grid(P, C, B, nullptr, nullptr, nx, name, &ncell);
    if (rc) return rc;
    AL3D_REQUIRE(out && workspace && x && (!lss || depth) && P > 0, "%s: null pointer", name);
    const BevPlan p = bev_plan_carve(workspace, P, ncell);
    const unsigned cblocks = (unsigned)al3d_cdiv(ncell, 4);
    if (C % 4 == 0 && C / 4 <= 64 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const int G = C / 4, CPW = 64 / G < BEV_CPW ? 64 / G : BEV_CPW;
        const unsigned vblocks = (unsigned)al3d_cdiv(ncell, 4 * CPW);
        if (lss)
            hipLaunchKernelGGL(bev_sum_vec_kernel<1>, dim3(vblocks), dim3(256), 0, s, x, depth, C, G, CPW, Dd, fHW, p.count, p.start,
                               p.list, p.sorted, ncell, out);
        else
            hipLaunchKernelGGL(bev_sum_vec_kernel<0>, dim3(vblocks), dim3(256), 0, s, x, depth, C, G, CPW, Dd, fHW, p.count, p.start,
                               p.list, p.sorted, ncell, out);
    } else if (lss)
        hipLaunchKernelGGL(bev_sum_kernel<1>, dim3(cblocks), dim3(256), 0, s, x, depth, C, Dd, fHW, p.count, p.start, p.list,
                           p.sorted, ncell, nx[2], out);
    else
        hipLaunchKernelGGL(bev_sum_kernel<0>, dim3(cblocks), dim3(256), 0, s, x, depth, C, Dd, fHW, p.count, p.start, p.list,
                           p.sorted, ncell, nx[2], out);
    AL3D_CHECK_LAUNCH(name);
    return AL3D_OK;
}

static int bev_pool_run(const float* x, const float* depth, int lss, int Dd, int fHW, const float* geom, int64_t P,
                        int C, int B, const float* lo, const float* dx, const int* nx, float* out, void* workspace,
                        hipStream_t s, const char* name)
{
    AL3D_REQUIRE(P >= 0 && P < (1LL << 31) && C >= 1 && B >= 1 && lo && dx && nx, "%s: bad arguments", name);
    if (P == 0) {
        int ncell;
        int rc = bev_check_grid(P, C, B, lo, dx, nx, name, &ncell);
        if (rc) return rc;
        AL3D_REQUIRE(out, "%s: null pointer", name);
        if (hipMemsetAsync(out, 0, (size_t)ncell * C * 4, s) != hipSuccess) return al3d_fail(AL3D_ELAUNCH, "%s: memset failed", name);
        return AL3D_OK;
    }
    int rc = bev_pool_plan(geom, P, B, lo, dx, nx, workspace, s, name);
    if (rc) return rc;
    return bev_pool_apply(x, depth, lss, Dd, fHW, P, C, B, nx, out, workspace, s, name);
}

// The two halves as entry points: the plan (cell of every frustum point, members of every cell in ascending point order)
// is a function of the geometry -- i.e. of the calibration matrices -- only; a sweep over a fixed rig builds it once and
// applies it to every batch's (depth, context) maps.  `workspace` (al3d_bev_pool_workspace_bytes) holds the plan.
extern "C" int al3d_bev_pool_plan(const float* geom, int64_t n_points, int B, const float* lo, const float* dx, const int* nx,
                                  void* workspace, void* stream)
{
    return bev_pool_plan(geom, n_points, B, lo, dx, nx, workspace, (hipStream_t)stream, "al3d_bev_pool_plan");
}

extern "C" int al3d_bev_pool_lss_apply_f32(const float* depth, const float* ctx, int BN, int D, int fH, int fW, int C, int B,
                                           const int* nx, const void* plan_workspace, float* out, void* stream)
{
    AL3D_REQUIRE(BN >= 1 && D >= 1 && fH >= 1 && fW >= 1 && B >= 1 && BN % B == 0, "al3d_bev_pool_lss_apply_f32: bad shape");
    const int64_t P = (int64_t)BN * D * fH * fW;
    return bev_pool_apply(ctx, depth, 1, D, fH * fW, P, C, B, nx, out, (void*)plan_workspace, (hipStream_t)stream,
                          "al3d_bev_pool_lss_apply_f32");
}

extern "C" int al3d_bev_pool_f32(const float* x, const float* geom, int64_t n_points, int C, int B, const float* lo,
                                 const float* dx, const int* nx, float* out, void* workspace, void* stream)
{
    return bev_pool_run(x, nullptr, 0, 1, 1, geom, n_points, C, B, lo, dx, nx, out, workspace, (hipStream_t)stream,
                        "al3d_bev_pool_f32");
}

extern "C" int al3d_bev_pool_lss_f32(const float* depth, const float* ctx, const float* geom, int BN, int D, int fH,
                                     int fW, int C, int B, const float* lo, const float* dx, const int* nx, float* out,
                                     void* workspace, void* stream)
{
    AL3D_REQUIRE(BN >= 1 && D >= 1 && fH >= 1 && fW >= 1 && B >= 1 && BN % B == 0, "al3d_bev_pool_lss_f32: bad shape");
    const int64_t P = (int64_t)BN * D * fH * fW;
    return bev_pool_run(ctx, depth, 1, D, fH * fW, geom, P, C, B, lo, dx, nx, out, workspace, (hipStream_t)stream,
                        "al3d_bev_pool_lss_f32");
}


// ---------------------------------------------------------------------------------------------
// Frustum geometry (vtransforms/base.py:79-122): for camera bn and frustum point f = (u, v, d):
//   q = inv(post_rot) (f - post_trans);  q = (q.x*q.z, q.y*q.z, q.z);  g = combine q + cam_trans
//   [g = extra_rot g] [g += extra_trans]       with combine = camera2lidar_rot inv(intrins).
// The 3x3 inverses / products are per-camera host algebra; this kernel does the 2 M points per sample that take
// torch's batched matmul 53 ms.  Dot products are evaluated as ((a0*x + a1*y) + a2*z), no contraction.
struct LssCam { float ipr[9], pt[3], comb[9], ct[3], er[9], et[3]; int has_er, has_et; };

__global__ void lss_geometry_kernel(const float* __restrict__ frustum, const LssCam* __restrict__ cams, int64_t per_cam,
                                    int64_t total, float* __restrict__ geom)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int bn = (int)(p / per_cam);
    const int64_t f = p % per_cam;
    const LssCam& c = cams[bn];
    const float x = frustum[3 * f] - c.pt[0], y = frustum[3 * f + 1] - c.pt[1], z = frustum[3 * f + 2] - c.pt[2];
    float qx = (c.ipr[0] * x + c.ipr[1] * y) + c.ipr[2] * z;
    float qy = (c.ipr[3] * x + c.ipr[4] * y) + c.ipr[5] * z;
    const float qz = (c.ipr[6] * x + c.ipr[7] * y) + c.ipr[8] * z;
    qx = qx * qz; qy = qy * qz;
    float gx = ((c.comb[0] * qx + c.comb[1] * qy) + c.comb[2] * qz) + c.ct[0];
    float gy = ((c.comb[3] * qx + c.comb[4] * qy) + c.comb[5] * qz) + c.ct[1];
    float gz = ((c.comb[6] * qx + c.comb[7] * qy) + c.comb[8] * qz) + c.ct[2];
    if (c.has_er) {
        const float rx = (c.er[0] * gx + c.er[1] * gy) + c.er[2] * gz;
        const float ry = (c.er[3] * gx + c.er[4] * gy) + c.er[5] * gz;
        const float rz = (c.er[6] * gx + c.er[7] * gy) + c.er[8] * gz;
        gx = rx; gy = ry; gz = rz;
    }
    if (c.has_et) { gx += c.et[0]; gy += c.et[1]; gz += c.et[2]; }
    geom[3 * p] = gx; geom[3 * p + 1] = gy; geom[3 * p + 2] = gz;
}

// cams: [BN][44] float32 rows = inv(post_rot)[9] | post_trans[3] | combine[9] | cam_trans[3] | extra_rot[9] |
// extra_trans[3] | has_extra_rot | has_extra_trans (the two flags as 0.0 / 1.0), device memory
__global__ void lss_unpack_cams_kernel(const float* __restrict__ rows, int BN, LssCam* __restrict__ cams)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BN) return;
    const float* r = rows + 44 * i;
    LssCam c;
    for (int k = 0; k < 9; ++k) { c.ipr[k] = r[k]; c.comb[k] = r[12 + k]; c.er[k] = r[24 + k]; }
    for (int k = 0; k < 3; ++k) { c.pt[k] = r[9 + k]; c.ct[k] = r[21 + k]; c.et[k] = r[33 + k]; }
    c.has_er = r[36] != 0.f; c.has_et = r[37] != 0.f;
    cams[i] = c;
}

extern "C" int64_t al3d_lss_geometry_workspace_bytes(int BN) { return al3d_align((int64_t)(BN > 0 ? BN : 1) * sizeof(LssCam), 256); }

extern "C" int al3d_lss_geometry_f32(const float* frustum, int64_t points_per_camera, const float* cam_rows, int BN,
                                     float* geom, void* workspace, void* stream)
{
    AL3D_REQUIRE(frustum && cam_rows && geom && workspace && BN >= 1 && points_per_camera >= 1,
                 "al3d_lss_geometry_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    LssCam* cams = (LssCam*)workspace;
    hipLaunchKernelGGL(lss_unpack_cams_kernel, dim3((unsigned)al3d_cdiv(BN, 64)), dim3(64), 0, s, cam_rows, BN, cams);
    const int64_t total = points_per_camera * BN;
    hipLaunchKernelGGL(lss_geometry_kernel, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, s, frustum, cams,
                       points_per_camera, total, geom);
    AL3D_CHECK_LAUNCH("lss_geometry_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ depth distribution of the depth net
// depth_lss.py:93-96: depth = x[:, :D].softmax(dim=1) on the depth net's output.  Here the net's map is channels-last
// y [BN][fH][fW][ldy] (D depth logits first, then the context channels) and the pooling wants probabilities as
// [BN][D][fH][fW]: one wave per pixel -- max, exp, sum over the D logits (D <= 256), normalise, transposed store.
__global__ __launch_bounds__(256) void lss_depth_softmax_kernel(const float* __restrict__ y, int64_t pixels, int hw, int D, int ldy,
                                                                float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (px >= pixels) return;                           // wave-uniform
    const float* row = y + px * ldy;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int d = lane + 64 * t;
        v[t] = d < D ? row[d] : -INFINITY;
        mx = fmaxf(mx, v[t]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { v[t] = expf(v[t] - mx); sum += v[t]; }      // exp(-inf) = 0 for the lanes beyond D
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const int64_t img = px / hw, pos = px - img * hw;
    float* o = out + img * (int64_t)D * hw + pos;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int d = lane + 64 * t;
        if (d < D) o[(int64_t)d * hw] = v[t] / sum;
    }
}

extern "C" int al3d_lss_depth_softmax_f32(const float* y, int BN, int fH, int fW, int D, int ldy, float* out, void* stream)
{
    AL3D_REQUIRE(y && out && BN >= 1 && fH >= 1 && fW >= 1 && D >= 1 && D <= 256 && ldy >= D,
                 "al3d_lss_depth_softmax_f32: bad arguments (1 <= D <= 256 <= ... ldy >= D)");
    const int64_t pixels = (int64_t)BN * fH * fW;
    hipLaunchKernelGGL(lss_depth_softmax_kernel, dim3((unsigned)al3d_cdiv(pixels, 4)), dim3(256), 0, (hipStream_t)stream, y, pixels,
                       fH * fW, D, ldy, out);
    AL3D_CHECK_LAUNCH("lss_depth_softmax_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ top-down step of the LSS-FPN
// generalized_lss.py:88-101: x = cat([laterals[i], interpolate(laterals[i + 1], size = laterals[i].shape, **upsample_cfg)],
// channel), upsample_cfg = {mode: bilinear, align_corners: ...} (the class defaults to True, the shipped swint configs set
// false: configs/nuscenes/det/transfusion/secfpn/camera+lidar/default.yaml:16-18).  One kernel writes the concatenated
// channels-last map: a thread owns four channels of one output pixel -- the first C1 are copied from `lat`, the other C2 are
// the four-tap blend of `src` with torch's weights (source position in float: align_corners: o (in - 1) / (out - 1);
// otherwise max(0, (o + 0.5) in / out - 0.5); the upper neighbour clamped at the border; blend as
// h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11)).
template <bool ALIGN>
__global__ __launch_bounds__(256) void lss_upsample_cat_kernel(const float* __restrict__ lat, const float* __restrict__ src, int64_t total,
                                                               int H, int W, int C1, int h, int w, int C2, float sh, float sw,
                                                               float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int CQ = (C1 + C2) >> 2;
    const int cq = (int)(t % CQ);
    const int64_t px = t / CQ;
    const int c = cq * 4;
    float4 v;
    if (c < C1) {
        v = *reinterpret_cast<const float4*>(lat + px * C1 + c);
    } else {
        const int x = (int)(px % W), y = (int)((px / W) % H);
        const int64_t n = px / ((int64_t)W * H);
        float yr, xr;
        if constexpr (ALIGN) {
            yr = sh * (float)y; xr = sw * (float)x;
        } else {                                                    // torch's area_pixel_compute_source_index, cubic = false
            yr = sh * ((float)y + 0.5f) - 0.5f; xr = sw * ((float)x + 0.5f) - 0.5f;
            yr = yr < 0.f ? 0.f : yr; xr = xr < 0.f ? 0.f : xr;
        }
        const int y0 = (int)yr, x0 = (int)xr;
        const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
        const float ly = yr - (float)y0, lx = xr - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float* b = src + ((n * h + y0) * w + x0) * C2 + (c - C1);
        const float4 v00 = *reinterpret_cast<const float4*>(b), v01 = *reinterpret_cast<const float4*>(b + (int64_t)xp * C2);
        const float4 v10 = *reinterpret_cast<const float4*>(b + (int64_t)yp * w * C2);
        const float4 v11 = *reinterpret_cast<const float4*>(b + ((int64_t)yp * w + xp) * C2);
        v.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        v.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        v.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        v.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    }
    *reinterpret_cast<float4*>(out + px * (C1 + C2) + c) = v;
}

extern "C" int al3d_lss_upsample_cat_mode_f32(const float* lat, const float* src, int N, int H, int W, int C1, int h, int w, int C2,
                                              int align_corners, float* out, void* stream);

extern "C" int al3d_lss_upsample_cat_f32(const float* lat, const float* src, int N, int H, int W, int C1, int h, int w, int C2,
                                         float* out, void* stream)
{
    return al3d_lss_upsample_cat_mode_f32(lat, src, N, H, W, C1, h, w, C2, 1, out, stream);
}

extern "C" int al3d_lss_upsample_cat_mode_f32(const float* lat, const float* src, int N, int H, int W, int C1, int h, int w, int C2,
                                              int align_corners, float* out, void* stream)
{
    AL3D_REQUIRE(lat && src && out, "al3d_lss_upsample_cat_f32: null pointer");
    AL3D_REQUIRE(N >= 0 && H >= 1 && W >= 1 && h >= 1 && w >= 1 && C1 >= 4 && C2 >= 4 && C1 % 4 == 0 && C2 % 4 == 0,
                 "al3d_lss_upsample_cat_f32: channel counts must be multiples of 4");
    AL3D_REQUIRE((((uintptr_t)lat | (uintptr_t)src | (uintptr_t)out) & 15) == 0, "al3d_lss_upsample_cat_f32: 16-byte aligned maps");
    const int64_t total = (int64_t)N * H * W * ((C1 + C2) / 4);
    if (total == 0) return AL3D_OK;
    if (align_corners) {
        const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
        hipLaunchKernelGGL(lss_upsample_cat_kernel<true>, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           lat, src, total, H, W, C1, h, w, C2, sh, sw, out);
    } else {
        const float sh = (float)h / (float)H, sw = (float)w / (float)W;        // area_pixel_compute_scale without scale factors
        hipLaunchKernelGGL(lss_upsample_cat_kernel<false>, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           lat, src, total, H, W, C1, h, w, C2, sh, sw, out);
    }
    AL3D_CHECK_LAUNCH("lss_upsample_cat_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ channel concatenation of two channels-last maps
// out [N][H][W][Ca + Cb] = cat(a, b) per pixel (depth_lss.py:84: cat([d, x], dim=1) before the depth net; fusers/conv.py:24:
// cat(inputs, dim=1) before the fuser's convolution).  a_hw_swapped: `a` is stored [N][W][H][Ca] -- the view transform's BEV
// map comes out as [x, y] and every other map of this build is [H = y, W = x]: the transposition rides on the copy instead of
// being a pass of its own.  One thread per four channels of an output pixel.
__global__ __launch_bounds__(256) void cat2_nhwc_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t total,
                                                        int H, int W, int Ca, int Cb, int swapped, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int CQ = (Ca + Cb) >> 2;
    const int c = (int)(t % CQ) * 4;
    const int64_t px = t / CQ;
    float4 v;
    if (c < Ca) {
        int64_t pa = px;
        if (swapped) {
            const int x = (int)(px % W), y = (int)((px / W) % H);
            const int64_t n = px / ((int64_t)W * H);
            pa = (n * W + x) * H + y;
        }
        v = *reinterpret_cast<const float4*>(a + pa * Ca + c);
    } else {
        v = *reinterpret_cast<const float4*>(b + px * Cb + (c - Ca));
    }
    *reinterpret_cast<float4*>(out + px * (Ca + Cb) + c) = v;
}

extern "C" int al3d_cat2_nhwc_f32(const float* a, const float* b, int N, int H, int W, int Ca, int Cb, int a_hw_swapped,
                                  float* out, void* stream)
{
    AL3D_REQUIRE(a && b && out, "al3d_cat2_nhwc_f32: null pointer");
    AL3D_REQUIRE(N >= 0 && H >= 1 && W >= 1 && Ca >= 4 && Cb >= 4 && Ca % 4 == 0 && Cb % 4 == 0,
                 "al3d_cat2_nhwc_f32: channel counts must be multiples of 4");
    AL3D_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "al3d_cat2_nhwc_f32: 16-byte aligned maps");
    const int64_t total = (int64_t)N * H * W * ((Ca + Cb) / 4);
    if (total == 0) return AL3D_OK;
    hipLaunchKernelGGL(cat2_nhwc_kernel, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, b, total, H, W,
                       Ca, Cb, a_hw_swapped ? 1 : 0, out);
    AL3D_CHECK_LAUNCH("cat2_nhwc_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ first two layers of the depth branch, fused
// depth_lss.py:38-44: dtransform = Conv2d(1, 8, 1) + BN + ReLU -> Conv2d(8, 32, 5, stride 4, padding 2) + BN + ReLU on
// the [BN, 1, iH, iW] lidar depth image.  As two convolutions that is a 554 MB eight-channel map per 16 samples written,
// padded to the matrix-core kernels' 16 input channels and read back; here one thread owns one output pixel and all C1
// output channels: the 1x1 layer is re-evaluated from the depth value of each of the 25 taps (4 flops per channel), the
// 5x5 weights are wave-uniform (scalar loads) and the accumulation is a plain fp32 FMA chain (taps, then channels).
// Zero padding applies to the FIRST layer's output (a tap outside the image contributes 0, not layer0(0)).
// p0 = [w0 | scale0 | shift0] (C0 each), w1 [KS*KS][C0][C1], p1 = [scale1 | shift1] (C1 each); out [BN][oH][oW][C1].
template <int C0, int C1, int KS, int ST, int PD>
__global__ __launch_bounds__(256) void lss_dtransform01_kernel(const float* __restrict__ depth, int64_t total, int iH, int iW,
                                                               int oH, int oW, const float* __restrict__ p0,
                                                               const float* __restrict__ w1, const float* __restrict__ p1,
                                                               float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < total;
    const int64_t tt = live ? t : 0;
    const int ox = (int)(tt % oW), oy = (int)((tt / oW) % oH);
    const int64_t n = tt / ((int64_t)oW * oH);
    const float* img = depth + n * iH * iW;
    float acc[C1];
#pragma unroll
    for (int co = 0; co < C1; ++co) acc[co] = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
        const int y = oy * ST - PD + ky;
        for (int kx = 0; kx < KS; ++kx) {
            const int x = ox * ST - PD + kx;
            const bool inb = y >= 0 && y < iH && x >= 0 && x < iW;
            const float d = inb ? img[(int64_t)y * iW + x] : 0.f;
            const float* w = w1 + (ky * KS + kx) * (C0 * C1);
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                float a = (p0[c] * d) * p0[C0 + c] + p0[2 * C0 + c];
                a = a <= 0.f ? 0.f : a;                                  // NaN propagates, like torch.relu
                a = inb ? a : 0.f;
#pragma unroll
                for (int co = 0; co < C1; ++co) acc[co] = __builtin_fmaf(w[c * C1 + co], a, acc[co]);
            }
        }
    }
    if (!live) return;
    float* o = out + t * C1;
#pragma unroll
    for (int co = 0; co < C1; co += 4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = acc[co + e] * p1[co + e] + p1[C1 + co + e];
            v[e] = v[e] <= 0.f ? 0.f : v[e];
        }
        *reinterpret_cast<float4*>(o + co) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int al3d_lss_dtransform01_f32(const float* depth, int BN, int iH, int iW, const float* p0, const float* w1,
                                         const float* p1, float* out, void* stream)
{
    AL3D_REQUIRE(depth && p0 && w1 && p1 && out, "al3d_lss_dtransform01_f32: null pointer");
    AL3D_REQUIRE(BN >= 0 && iH >= 1 && iW >= 1, "al3d_lss_dtransform01_f32: bad shape");
    AL3D_REQUIRE(((uintptr_t)out & 15) == 0, "al3d_lss_dtransform01_f32: out must be 16-byte aligned");
    const int oH = (iH + 2 * 2 - 5) / 4 + 1, oW = (iW + 2 * 2 - 5) / 4 + 1;
    const int64_t total = (int64_t)BN * oH * oW;
    if (total == 0) return AL3D_OK;
    hipLaunchKernelGGL((lss_dtransform01_kernel<8, 32, 5, 4, 2>), dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, depth, total, iH, iW, oH, oW, p0, w1, p1, out);
    AL3D_CHECK_LAUNCH("lss_dtransform01_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ lidar depth image of the depth-aware LSS transform
// BaseDepthTransform.forward (bevfusion/mmdet3d/models/vtransforms/base.py:225-262): every lidar point of a sample is
// taken back through the lidar augmentation, projected into each camera (lidar2image), through the image
// augmentation, truncated to a pixel, and its depth written to depth[b, cam, 0, row, col]; points that land on one
// pixel overwrite each other in point order (the reference's indexed assignment on the CPU keeps the LAST point).
// One thread per (camera, point); the overwrite order is made deterministic with a 64-bit atomicMax on
// (point index + 1) << 32 | depth bits, resolved by a second pass.  Dot products as ((a0 x + a1 y) + a2 z), no
// contraction.  cam_rows [N][24]: lidar2image[:3,:3] (9) | lidar2image[:3,3] (3) | img_aug[:3,:3] (9) | img_aug[:3,3] (3);
// aug [12]: inverse(lidar_aug[:3,:3]) (9) | lidar_aug[:3,3] (3).
__global__ void lss_depth_scatter_kernel(const float* __restrict__ pts, int64_t npts, int stride, const float* __restrict__ cam_rows,
                                         int ncam, const float* __restrict__ aug, int iH, int iW,
                                         unsigned long long* __restrict__ keys)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npts * ncam) return;
    const int c = (int)(t / npts);
    const int64_t i = t % npts;
    const float* p = pts + i * stride;
    float x = p[0] - aug[9], y = p[1] - aug[10], z = p[2] - aug[11];
    const float ux = (aug[0] * x + aug[1] * y) + aug[2] * z;
    const float uy = (aug[3] * x + aug[4] * y) + aug[5] * z;
    const float uz = (aug[6] * x + aug[7] * y) + aug[8] * z;
    const float* m = cam_rows + 24 * c;
    float qx = ((m[0] * ux + m[1] * uy) + m[2] * uz) + m[9];
    float qy = ((m[3] * ux + m[4] * uy) + m[5] * uz) + m[10];
    float qz = ((m[6] * ux + m[7] * uy) + m[8] * uz) + m[11];
    qz = fminf(fmaxf(qz, 1e-5f), 1e5f);
    const float dist = qz;     // base.py:236-237: `dist` is a VIEW of the row the clamp then writes in place -> the clamped depth
    qx = qx / qz; qy = qy / qz;
    const float* a = m + 12;
    const float vx = ((a[0] * qx + a[1] * qy) + a[2] * qz) + a[9];
    const float vy = ((a[3] * qx + a[4] * qy) + a[5] * qz) + a[10];
    // the reference swaps to (row, col) = (vy, vx) and keeps 0 <= row < iH, 0 <= col < iW, then truncates
    if (!(vy < (float)iH && vy >= 0.f && vx < (float)iW && vx >= 0.f)) return;
    const int row = (int)vy, col = (int)vx;
    const unsigned long long key = ((unsigned long long)(i + 1) << 32) | (unsigned long long)__float_as_uint(dist);
    atomicMax(&keys[((int64_t)c * iH + row) * iW + col], key);
}

__global__ void lss_depth_resolve_kernel(const unsigned long long* __restrict__ keys, int64_t n, float* __restrict__ depth)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned long long k = keys[t];
    depth[t] = k ? __uint_as_float((unsigned)(k & 0xffffffffull)) : 0.f;
}

extern "C" int64_t al3d_lss_depth_image_workspace_bytes(int ncam, int iH, int iW)
{
    return al3d_align((int64_t)(ncam > 0 ? ncam : 1) * iH * iW * 8, 256);
}

// depth [ncam][iH][iW] of ONE sample; points [npts][stride >= 3] f32
extern "C" int al3d_lss_depth_image_f32(const float* points, int64_t npts, int stride, const float* cam_rows, int ncam,
                                        const float* aug_rows, int iH, int iW, float* depth, void* workspace, void* stream)
{
    AL3D_REQUIRE(cam_rows && aug_rows && depth && workspace && ncam >= 1 && iH >= 1 && iW >= 1 && npts >= 0 && stride >= 3,
                 "al3d_lss_depth_image_f32: bad arguments");
    AL3D_REQUIRE(npts == 0 || points, "al3d_lss_depth_image_f32: null points");
    AL3D_REQUIRE(npts < ((int64_t)1 << 31), "al3d_lss_depth_image_f32: too many points");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)ncam * iH * iW;
    if (hipMemsetAsync(workspace, 0, (size_t)n * 8, s) != hipSuccess) return al3d_fail(AL3D_ELAUNCH, "al3d_lss_depth_image_f32: memset failed");
    if (npts > 0) {
        hipLaunchKernelGGL(lss_depth_scatter_kernel, dim3((unsigned)al3d_cdiv(npts * ncam, 256)), dim3(256), 0, s, points, npts,
                           stride, cam_rows, ncam, aug_rows, iH, iW, (unsigned long long*)workspace);
        AL3D_CHECK_LAUNCH("lss_depth_scatter_kernel");
    }
    hipLaunchKernelGGL(lss_depth_resolve_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0, s,
                       (const unsigned long long*)workspace, n, depth);
    AL3D_CHECK_LAUNCH("lss_depth_resolve_kernel");
    return AL3D_OK;
}
