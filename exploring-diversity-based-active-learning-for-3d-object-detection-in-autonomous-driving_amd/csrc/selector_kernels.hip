// Selector kernels for gfx950: pairwise L1 embedding distance, distance-map
// normalise+aggregate, max-finite reduction and the persistent greedy k-center
// loop.  Built with -ffp-contract=off: every rounding below is the one written.
//
// Reference semantics (paths relative to the reference repo):
//   L1 map      det3d/selectors/feature_selector.py:87-109
//   combine     det3d/selectors/spatial_temporal_selector.py:109-155
//               det3d/selectors/spatial_temporal_feature_selector.py:187-219
//   greedy      det3d/selectors/spatial_temporal_selector.py:157-193
//               det3d/selectors/feature_selector.py:142-172
#include "al3d_common.h"
#include "al3d_exp.h"

thread_local char g_al3d_err[512] = "";

extern "C" int al3d_abi_version(void) { return 1; }
extern "C" const char* al3d_last_error(void) { return g_al3d_err; }

// ------------------------------------------------------------------ L1 map
// 64x64 output tile per 256-thread workgroup, 4x4 outputs per thread, feats
// staged k-major in LDS so each thread reads its 4 rows / 4 cols as one
// ds_read_b128 each.  VALU-bound (sub + |.|-add per element pair), not MFMA:
// an absolute difference is not a contraction.
#define L1_TILE 64
#define L1_KC 32

template <int P>
__device__ __forceinline__ float l1_term(float d)
{
    if (P == 1) return fabsf(d);
    // sqrt(d*d) == |d| whenever d*d neither underflows nor overflows (binary
    // round-to-nearest); evaluate the literal expression outside that range.
    float a = fabsf(d);
    return (a >= 0x1p-60f && a <= 0x1p60f) ? a : sqrtf(d * d);
}

template <int P>
__global__ __launch_bounds__(256) void l1_map_kernel(const float* __restrict__ feats, int64_t n,
                                                     int64_t c, int64_t row0, int64_t nrows,
                                                     float* __restrict__ out)   // out: rows [row0, row0+nrows)
{
    __shared__ __attribute__((aligned(16))) float As[L1_KC][L1_TILE];
    __shared__ __attribute__((aligned(16))) float Bs[L1_KC][L1_TILE];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = row0 + (int64_t)blockIdx.y * L1_TILE;  // rows: feats[i]
    const int64_t iend = row0 + nrows;
    const int64_t j0 = (int64_t)blockIdx.x * L1_TILE;  // cols: feats[j]
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[r][s] = 0.0f;

    for (int64_t k0 = 0; k0 < c; k0 += L1_KC) {
        // stage: 64 rows x 32 k per operand; lane -> consecutive k (128-B row segments)
#pragma unroll
        for (int t = 0; t < (L1_TILE * L1_KC) / 256; ++t) {
            int e = tid + t * 256;
            int row = e / L1_KC, kk = e % L1_KC;
            int64_t k = k0 + kk;
            float av = 0.0f, bv = 0.0f;
            if (k < c) {
                if (i0 + row < n) av = feats[(i0 + row) * c + k];
                if (j0 + row < n) bv = feats[(j0 + row) * c + k];
            }
            As[kk][row] = av;
            Bs[kk][row] = bv;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < L1_KC; ++kk) {
            float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            float4 b4 = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            float a[4] = {a4.x, a4.y, a4.z, a4.w};
            float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[r][s] += l1_term<P>(b[s] - a[r]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int64_t i = i0 + ty * 4 + r;
        if (i >= n || i >= iend) continue;
        int64_t j = j0 + tx * 4;
        float* orow = out + (i - row0) * n;
        if (j + 3 < n && (n & 3) == 0) {
            *reinterpret_cast<float4*>(&orow[j]) =
                make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (j + s < n) orow[j + s] = acc[r][s];
        }
    }
}

extern "C" int al3d_l1_distance_rows_f32(const float* feats, int64_t n, int64_t c, int p, int64_t row0,
                                         int64_t nrows, float* out, void* stream)
{
    AL3D_REQUIRE(n >= 0 && c >= 0, "al3d_l1_distance_rows_f32: negative size");
    AL3D_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= n, "al3d_l1_distance_rows_f32: bad row range");
    AL3D_REQUIRE(p == 1 || p == 2, "al3d_l1_distance_rows_f32: p must be 1 or 2 (got %d)", p);
    if (n == 0 || nrows == 0) return AL3D_OK;
    AL3D_REQUIRE(feats && out, "al3d_l1_distance_rows_f32: null pointer");
    AL3D_REQUIRE(((uintptr_t)out & 15) == 0, "al3d_l1_distance_rows_f32: out must be 16-byte aligned");
    dim3 grid((unsigned)al3d_cdiv(n, L1_TILE), (unsigned)al3d_cdiv(nrows, L1_TILE));
    hipStream_t s = (hipStream_t)stream;
    if (p == 1) hipLaunchKernelGGL(l1_map_kernel<1>, grid, dim3(256), 0, s, feats, n, c, row0, nrows, out);
    else hipLaunchKernelGGL(l1_map_kernel<2>, grid, dim3(256), 0, s, feats, n, c, row0, nrows, out);
    AL3D_CHECK_LAUNCH("l1_map_kernel");
    return AL3D_OK;
}

extern "C" int al3d_l1_distance_f32(const float* feats, int64_t n, int64_t c, int p, float* out,
                                    void* stream)
{
    AL3D_REQUIRE(n >= 0 && c >= 0, "al3d_l1_distance_f32: negative size");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(feats && out, "al3d_l1_distance_f32: null pointer");
    return al3d_l1_distance_rows_f32(feats, n, c, p, 0, n, out, stream);
}

// ------------------------------------------------------------------ combine
// One pass over the [n,n] maps: HBM-bound (reads 8 B spatial [+4 B feature],
// writes 8 B per element); the temporal term is recomputed from the ids.
__device__ __forceinline__ double temporal_gap(const int64_t* __restrict__ id, int64_t i,
                                               int64_t idi, int64_t j)
{
    if (id[j] != idi) return 1e6;
    return (double)(i > j ? i - j : j - i);
}

__global__ __launch_bounds__(256) void combine_kernel(
    const double* __restrict__ spatial, const int64_t* __restrict__ tid_, const float* __restrict__ feat,
    int64_t n, int normalize, int aggregate, double lambda_t, float lambda_f_f32,
    double spatial_scale, double temporal_scale, double* __restrict__ out)
{
    const int64_t i = blockIdx.y;
    const int64_t idi = tid_ ? tid_[i] : 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * n + j;
        double s = 0.0, t = 0.0, f = 0.0;
        if (spatial) {
            s = spatial[e];
            if (normalize == AL3D_NORM_EXP) s = 1.0 - al3d_exp_f64(-s);
            else if (normalize == AL3D_NORM_LINEAR) s = s / spatial_scale;
        }
        if (tid_) {
            t = temporal_gap(tid_, i, idi, j);
            if (normalize == AL3D_NORM_EXP) t = 1.0 - al3d_exp_f64(-t);
            else if (normalize == AL3D_NORM_LINEAR) t = t / temporal_scale;
        }
        if (feat) {
            float ff = feat[e];
            if (normalize == AL3D_NORM_EXP) ff = 1.0f - al3d_exp_f32(-ff);
            if (aggregate == AL3D_AGG_SUM) ff = lambda_f_f32 * ff;
            f = (double)ff;
        }
        double r;
        if (aggregate == AL3D_AGG_SUM) {
            r = spatial ? s : 0.0;
            if (tid_) r = spatial ? r + lambda_t * t : lambda_t * t;
            if (feat) r = (spatial || tid_) ? r + f : f;
        } else {
            bool have = false;
            r = 0.0;
            if (spatial) { r = s; have = true; }
            if (tid_) { r = !have ? t : (aggregate == AL3D_AGG_MIN ? (t < r ? t : r) : (t > r ? t : r)); have = true; }
            if (feat) { r = !have ? f : (aggregate == AL3D_AGG_MIN ? (f < r ? f : r) : (f > r ? f : r)); }
        }
        out[e] = r;
    }
}

extern "C" int al3d_combine_maps_f64(const double* spatial, const int64_t* temporal_id,
                                     const float* feat, int64_t n, int normalize, int aggregate,
                                     double lambda_t, double lambda_f, double spatial_scale,
                                     double temporal_scale, double* out, void* stream)
{
    AL3D_REQUIRE(out, "al3d_combine_maps_f64: null output");
    AL3D_REQUIRE(spatial || temporal_id || feat, "al3d_combine_maps_f64: no input term");
    AL3D_REQUIRE(normalize >= 0 && normalize <= 2, "al3d_combine_maps_f64: bad normalize %d", normalize);
    AL3D_REQUIRE(aggregate >= 0 && aggregate <= 2, "al3d_combine_maps_f64: bad aggregate %d", aggregate);
    AL3D_REQUIRE(n >= 0 && n < (1LL << 31), "al3d_combine_maps_f64: bad n");
    if (n == 0) return AL3D_OK;
    unsigned gx = (unsigned)al3d_cdiv(n, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       spatial, temporal_id, feat, n, normalize, aggregate, lambda_t,
                       (float)lambda_f, spatial_scale, temporal_scale, out);
    AL3D_CHECK_LAUNCH("combine_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ euclid map
// EuSpatialSelector (det3d/selectors/euclidean_spatial_selector.py:95-106):
// D[i][j] = sqrt(dx^2 + dy^2) inside one map location, 1e6 across locations.
__global__ __launch_bounds__(256) void euclid_map_kernel(const double* __restrict__ xy,
                                                         const int64_t* __restrict__ loc_id, int64_t n,
                                                         double* __restrict__ out)
{
    const int64_t i = blockIdx.y;
    const double xi = xy[2 * i], yi = xy[2 * i + 1];
    const int64_t li = loc_id[i];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        double dx = xy[2 * j] - xi, dy = xy[2 * j + 1] - yi;
        double d2 = dx * dx;
        double dy2 = dy * dy;
        d2 = d2 + dy2;
        out[i * n + j] = loc_id[j] == li ? sqrt(d2) : 1e6;
    }
}

extern "C" int al3d_euclid_map_f64(const double* xy, const int64_t* loc_id, int64_t n, double* out,
                                   void* stream)
{
    AL3D_REQUIRE(xy && loc_id && out, "al3d_euclid_map_f64: null pointer");
    AL3D_REQUIRE(n >= 0 && n < (1LL << 31), "al3d_euclid_map_f64: bad n");
    if (n == 0) return AL3D_OK;
    unsigned gx = (unsigned)al3d_cdiv(n, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(euclid_map_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, xy,
                       loc_id, n, out);
    AL3D_CHECK_LAUNCH("euclid_map_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ max finite
__global__ void max_finite_init(double* out) { *out = -__builtin_inf(); }

__global__ __launch_bounds__(256) void max_finite_kernel(const double* __restrict__ a, int64_t count,
                                                         double* out)
{
    double m = -__builtin_inf();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        double v = a[i];
        if (v != __builtin_inf() && v > m) m = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(m, off);
        if (o > m) m = o;
    }
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (part[w] > m) m = part[w];
        // non-negative maps: the f64 bit pattern orders like a signed integer
        if (m >= 0.0) atomicMax(reinterpret_cast<long long*>(out), __double_as_longlong(m));
    }
}

extern "C" int al3d_max_finite_f64(const double* a, int64_t count, double* out_dev, void* stream)
{
    AL3D_REQUIRE(a && out_dev, "al3d_max_finite_f64: null pointer");
    AL3D_REQUIRE(count >= 0, "al3d_max_finite_f64: negative count");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(max_finite_init, dim3(1), dim3(1), 0, s, out_dev);
    if (count > 0) {
        unsigned g = (unsigned)al3d_cdiv(count, 256 * 8);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(max_finite_kernel, dim3(g), dim3(256), 0, s, a, count, out_dev);
    }
    AL3D_CHECK_LAUNCH("max_finite_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ greedy
// The pick loop is a chain of dependent row reads (row address = previous
// argmax), so one persistent 1024-thread workgroup runs all of it: per pick one
// coalesced pass over D[last] fused with the fps min-update and a first-index
// argmax (wave shuffle -> LDS -> wave 0), two barriers, no host round trip.
// fps lives in a global scratch row each thread owns privately.
#define GREEDY_THREADS 1024

template <typename T>
struct GreedyBest { T v; int64_t i; };

template <typename T>
__device__ __forceinline__ void greedy_take(GreedyBest<T>& b, T ov, int64_t oi)
{
    // larger value wins; equal values -> lower index (np.argmax / torch.argmax first hit)
    if (oi >= 0 && (b.i < 0 || ov > b.v || (ov == b.v && oi < b.i))) { b.v = ov; b.i = oi; }
}

template <typename T>
__device__ __forceinline__ GreedyBest<T> greedy_block_argmax(GreedyBest<T> b, T* s_v, int64_t* s_i)
{
    for (int off = 32; off > 0; off >>= 1) {
        T ov = __shfl_down(b.v, off);
        int64_t oi = __shfl_down(b.i, off);
        greedy_take(b, ov, oi);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_v[wave] = b.v; s_i[wave] = b.i; }
    __syncthreads();
    if (wave == 0) {
        GreedyBest<T> w;
        w.i = lane < GREEDY_THREADS / 64 ? s_i[lane] : -1;
        w.v = lane < GREEDY_THREADS / 64 ? s_v[lane] : (T)0;
        for (int off = 8; off > 0; off >>= 1) {
            T ov = __shfl_down(w.v, off);
            int64_t oi = __shfl_down(w.i, off);
            greedy_take(w, ov, oi);
        }
        b = w;
    }
    return b;  // valid in thread 0
}

template <typename T>
__global__ __launch_bounds__(GREEDY_THREADS) void greedy_kernel(
    const T* __restrict__ D, const T* __restrict__ seed_map, int64_t n,
    const int64_t* __restrict__ seeded, int64_t n_seeded, int64_t first,
    const double* __restrict__ box_cost, double cost_f, double start_cost, double budget_int,
    int check_seeded, int64_t* __restrict__ out_idx, int64_t cap, int64_t* __restrict__ out_meta,
    T* __restrict__ fps, unsigned char* __restrict__ flags)
{
    __shared__ T s_v[GREEDY_THREADS / 64];
    __shared__ int64_t s_i[GREEDY_THREADS / 64];
    __shared__ int64_t s_sel;
    __shared__ int s_stop;
    const int tid = threadIdx.x;

    // flags[j]: bit0 = picked in this call, bit1 = member of the seeded list
    for (int64_t j = tid; j < n; j += GREEDY_THREADS) flags[j] = 0;
    __syncthreads();
    for (int64_t s = tid; s < n_seeded; s += GREEDY_THREADS) flags[seeded[s]] = 2;
    __syncthreads();

    GreedyBest<T> best;
    best.i = -1; best.v = (T)0;
    if (n_seeded > 0) {
        // fps = column-wise min over the seeded rows, first pick = argmax
        for (int64_t j = tid; j < n; j += GREEDY_THREADS) {
            T m = seed_map[seeded[0] * n + j];
            for (int64_t s = 1; s < n_seeded; ++s) {
                T v = seed_map[seeded[s] * n + j];
                if (v < m) m = v;
            }
            fps[j] = m;
            if (best.i < 0 || m > best.v) { best.v = m; best.i = j; }
        }
        best = greedy_block_argmax(best, s_v, s_i);
        if (tid == 0) s_sel = best.i;
    } else {
        for (int64_t j = tid; j < n; j += GREEDY_THREADS) fps[j] = seed_map[first * n + j];
        if (tid == 0) s_sel = first;
    }
    double cost = 0.0;
    int64_t cnt = 0;
    if (tid == 0) s_stop = 0;
    __syncthreads();
    if (tid == 0) {
        int64_t sel = s_sel;
        cost = start_cost;
        cost += cost_f;
        cost += box_cost[sel];
        out_idx[0] = sel;
        cnt = 1;
        flags[sel] |= 1;
    }
    int status = AL3D_GREEDY_OK;
    for (;;) {
        const int64_t sel = s_sel;
        const T* __restrict__ row = D + sel * n;
        best.i = -1; best.v = (T)0;
        for (int64_t j = tid; j < n; j += GREEDY_THREADS) {
            T f = fps[j];
            T r = row[j];
            if (r < f) f = r;
            fps[j] = f;
            if (best.i < 0 || f > best.v) { best.v = f; best.i = j; }
        }
        best = greedy_block_argmax(best, s_v, s_i);
        if (tid == 0) {
            const int64_t b = best.i;
            const unsigned char fl = flags[b];
            if ((fl & 1) || (check_seeded && (fl & 2))) { status = AL3D_GREEDY_DUPLICATE; s_stop = 1; }
            else {
                cost += cost_f;
                cost += box_cost[b];
                if (cost > budget_int) s_stop = 1;
                else if (cnt >= cap) { status = AL3D_GREEDY_FULL; s_stop = 1; }
                else { out_idx[cnt++] = b; flags[b] = fl | 1; s_sel = b; }
            }
        }
        __syncthreads();
        if (s_stop) break;
    }
    if (tid == 0) { out_meta[0] = cnt; out_meta[1] = status; }
}

extern "C" int64_t al3d_greedy_workspace_bytes(int64_t n, int elem_size)
{
    return al3d_align(n * (int64_t)elem_size, 256) + al3d_align(n, 256);
}

template <typename T>
static int greedy_launch(const T* D, const T* seed_map, int64_t n, const int64_t* seeded,
                         int64_t n_seeded, int64_t first, const double* box_cost, double cost_f,
                         double start_cost, double budget_int, int check_seeded, int64_t* out_idx,
                         int64_t cap, int64_t* out_meta, void* workspace, void* stream)
{
    AL3D_REQUIRE(D && seed_map && box_cost && out_idx && out_meta && workspace,
                 "al3d_greedy_kcenter: null pointer");
    AL3D_REQUIRE(n >= 1, "al3d_greedy_kcenter: empty pool");
    AL3D_REQUIRE(cap >= 1, "al3d_greedy_kcenter: cap must be >= 1");
    AL3D_REQUIRE(n_seeded >= 0, "al3d_greedy_kcenter: negative n_seeded");
    AL3D_REQUIRE(n_seeded == 0 || seeded, "al3d_greedy_kcenter: seeded list missing");
    AL3D_REQUIRE(n_seeded > 0 || (first >= 0 && first < n),
                 "al3d_greedy_kcenter: first pick %lld outside [0,%lld)", (long long)first, (long long)n);
    T* fps = reinterpret_cast<T*>(workspace);
    unsigned char* flags = reinterpret_cast<unsigned char*>(workspace) + al3d_align(n * (int64_t)sizeof(T), 256);
    hipLaunchKernelGGL(greedy_kernel<T>, dim3(1), dim3(GREEDY_THREADS), 0, (hipStream_t)stream, D,
                       seed_map, n, seeded, n_seeded, first, box_cost, cost_f, start_cost, budget_int,
                       check_seeded, out_idx, cap, out_meta, fps, flags);
    AL3D_CHECK_LAUNCH("greedy_kernel");
    return AL3D_OK;
}

extern "C" int al3d_greedy_kcenter_f64(const double* D, const double* seed_map, int64_t n,
                                       const int64_t* seeded, int64_t n_seeded, int64_t first,
                                       const double* box_cost, double cost_f, double start_cost,
                                       double budget_int, int check_seeded, int64_t* out_idx,
                                       int64_t cap, int64_t* out_meta, void* workspace, void* stream)
{
    return greedy_launch<double>(D, seed_map, n, seeded, n_seeded, first, box_cost, cost_f, start_cost,
                                 budget_int, check_seeded, out_idx, cap, out_meta, workspace, stream);
}

extern "C" int al3d_greedy_kcenter_f32(const float* D, const float* seed_map, int64_t n,
                                       const int64_t* seeded, int64_t n_seeded, int64_t first,
                                       const double* box_cost, double cost_f, double start_cost,
                                       double budget_int, int check_seeded, int64_t* out_idx,
                                       int64_t cap, int64_t* out_meta, void* workspace, void* stream)
{
    return greedy_launch<float>(D, seed_map, n, seeded, n_seeded, first, box_cost, cost_f, start_cost,
                                budget_int, check_seeded, out_idx, cap, out_meta, workspace, stream);
}
