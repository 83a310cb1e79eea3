// Spatial term on device: exact 2-D kNN, symmetric kNN-graph adjacency, and
// all-pairs shortest paths as N independent label-correcting sweeps (one
// workgroup per source, distances resident in LDS).
//
// Reference: det3d/selectors/spatial_temporal_selector.py:92-104
//   tree = cKDTree(locations); knn_d, knn_i = tree.query(locations, k+1)
//   W[a, knn_i[a]] = W[knn_i[a], a] = knn_d[a]           (0 == no edge)
//   shortest_path(W, directed=False, method="D")
//
// Bit-exactness: Dijkstra's result is the least fixed point of
// dist[v] = min_u fl(dist[u] + w(u,v)) with dist[src] = 0; because fl(a+w) is
// monotone in a and >= a for w > 0, any label-correcting iteration that only
// ever applies fl(dist[u] + w) and stops when no edge can lower a label lands
// on the same bits, whatever the relaxation order.  Built with
// -ffp-contract=off (dx*dx + dy*dy must round twice, like the reference).
#include "al3d_common.h"

// ------------------------------------------------------------------ kNN
// One thread per query point; candidate points stream through LDS in tiles;
// each thread keeps its kq best (squared distance, index) pairs in LDS columns
// (t-major, so lanes hit consecutive banks) and the current worst in a register.
#define KNN_THREADS 256

__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const double* __restrict__ xy, int64_t n,
                                                          int kq, double* __restrict__ knn_d,
                                                          int64_t* __restrict__ knn_i)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);                 // [KNN_THREADS][2]
    double* best_d = tile + 2 * KNN_THREADS;                         // [kq][KNN_THREADS]
    int* best_i = reinterpret_cast<int*>(best_d + (size_t)kq * KNN_THREADS);  // [kq][KNN_THREADS]
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * KNN_THREADS + tid;
    const bool live = q < n;
    const double qx = live ? xy[2 * q] : 0.0, qy = live ? xy[2 * q + 1] : 0.0;
    int cnt = 0;
    double worst = __builtin_inf();
    for (int64_t j0 = 0; j0 < n; j0 += KNN_THREADS) {
        __syncthreads();
        if (j0 + tid < n) {
            tile[2 * tid] = xy[2 * (j0 + tid)];
            tile[2 * tid + 1] = xy[2 * (j0 + tid) + 1];
        }
        __syncthreads();
        const int lim = (int)((n - j0) < KNN_THREADS ? (n - j0) : KNN_THREADS);
        if (!live) continue;
        for (int jj = 0; jj < lim; ++jj) {
            double dx = tile[2 * jj] - qx, dy = tile[2 * jj + 1] - qy;
            double d2 = dx * dx;
            d2 += dy * dy;
            if (cnt == kq && !(d2 < worst)) continue;
            int pos = cnt < kq ? cnt : kq - 1;
            while (pos > 0 && best_d[(size_t)(pos - 1) * KNN_THREADS + tid] > d2) {
                best_d[(size_t)pos * KNN_THREADS + tid] = best_d[(size_t)(pos - 1) * KNN_THREADS + tid];
                best_i[(size_t)pos * KNN_THREADS + tid] = best_i[(size_t)(pos - 1) * KNN_THREADS + tid];
                --pos;
            }
            best_d[(size_t)pos * KNN_THREADS + tid] = d2;
            best_i[(size_t)pos * KNN_THREADS + tid] = (int)(j0 + jj);
            if (cnt < kq) ++cnt;
            if (cnt == kq) worst = best_d[(size_t)(kq - 1) * KNN_THREADS + tid];
        }
    }
    if (!live) return;
    for (int t = 0; t < kq; ++t) {
        if (t < cnt) {
            knn_d[q * kq + t] = sqrt(best_d[(size_t)t * KNN_THREADS + tid]);
            knn_i[q * kq + t] = best_i[(size_t)t * KNN_THREADS + tid];
        } else {  // scipy pads missing neighbours with (inf, n)
            knn_d[q * kq + t] = __builtin_inf();
            knn_i[q * kq + t] = n;
        }
    }
}

extern "C" int al3d_knn_2d_f64(const double* xy, int64_t n, int kq, double* knn_d, int64_t* knn_i,
                               void* stream)
{
    AL3D_REQUIRE(xy && knn_d && knn_i, "al3d_knn_2d_f64: null pointer");
    AL3D_REQUIRE(kq >= 1 && kq <= 32, "al3d_knn_2d_f64: kq must be in [1,32] (got %d)", kq);
    AL3D_REQUIRE(n >= 0 && n < (1LL << 31), "al3d_knn_2d_f64: bad n");
    if (n == 0) return AL3D_OK;
    size_t lds = sizeof(double) * 2 * KNN_THREADS + (size_t)kq * KNN_THREADS * (sizeof(double) + sizeof(int));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(knn_kernel, dim3((unsigned)al3d_cdiv(n, KNN_THREADS)), dim3(KNN_THREADS), lds,
                       (hipStream_t)stream, xy, n, kq, knn_d, knn_i);
    AL3D_CHECK_LAUNCH("knn_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ adjacency
// Both directions of every kNN pair (duplicates kept: same weight bits, they
// only repeat a relaxation).  Zero-length and padded entries are dropped.
__device__ __forceinline__ bool edge_ok(int64_t b, double d, int64_t n)
{
    return b < n && d != 0.0 && d < __builtin_inf();
}

__global__ void adj_count_kernel(const double* __restrict__ knn_d, const int64_t* __restrict__ knn_i,
                                 int64_t n, int kq, int* __restrict__ deg)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * kq) return;
    int64_t a = e / kq, b = knn_i[e];
    if (!edge_ok(b, knn_d[e], n)) return;
    atomicAdd(&deg[a], 1);
    atomicAdd(&deg[b], 1);
}

// exclusive scan of deg[0..n) -> indptr[0..n], cursor copy; one workgroup.
__global__ __launch_bounds__(1024) void adj_scan_kernel(const int* __restrict__ deg, int64_t n,
                                                        int* __restrict__ indptr, int* __restrict__ cursor)
{
    __shared__ int s_part[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n; c0 += 1024) {
        int v = (c0 + tid < n) ? deg[c0 + tid] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        int excl = s_base + s_part[tid] - v;
        if (c0 + tid < n) { indptr[c0 + tid] = excl; cursor[c0 + tid] = excl; }
        __syncthreads();
        if (tid == 1023) s_base += s_part[1023];
        __syncthreads();
    }
    if (tid == 0) indptr[n] = s_base;
}

__global__ void adj_fill_kernel(const double* __restrict__ knn_d, const int64_t* __restrict__ knn_i,
                                int64_t n, int kq, int* __restrict__ cursor, int* __restrict__ adj_v,
                                double* __restrict__ adj_w)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * kq) return;
    int64_t a = e / kq, b = knn_i[e];
    double d = knn_d[e];
    if (!edge_ok(b, d, n)) return;
    int p = atomicAdd(&cursor[a], 1);
    adj_v[p] = (int)b; adj_w[p] = d;
    p = atomicAdd(&cursor[b], 1);
    adj_v[p] = (int)a; adj_w[p] = d;
}

// ------------------------------------------------------------------ SSSP sweeps
// One workgroup per source.  Labels are the f64 bit patterns (non-negative, so
// unsigned integer order == numeric order) lowered with integer atomic-min.
// Per round: the frontier flags (one BIT per node: a thread scans whole 32-node words) are
// compacted into a 16-bit node list, then 16-lane groups relax the edges of one
// frontier node each (edge-parallel: the dependent atomic round trips of a node's ~16
// edges overlap instead of queueing in one lane).  Two barriers per round.
// LDS_DIST: labels live in LDS (10.25 B/node incl. flags + list: up to 15.9k nodes); otherwise in
// the output row itself (global atomics at L2), 2.25 B/node of LDS (two workgroups per CU at 28k nodes).
#define SSSP_THREADS 256

template <bool LDS_DIST>
__global__ __launch_bounds__(SSSP_THREADS) void sssp_kernel(const int* __restrict__ indptr,
                                                            const int* __restrict__ adj_v,
                                                            const double* __restrict__ adj_w,
                                                            int64_t n, int64_t row0,
                                                            double* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int64_t src = row0 + blockIdx.x;
    double* orow = out + (int64_t)blockIdx.x * n;
    const int words = (int)((n + 31) >> 5);
    unsigned long long* dist;
    unsigned* fl;
    if (LDS_DIST) {
        dist = reinterpret_cast<unsigned long long*>(smem);
        fl = reinterpret_cast<unsigned*>(smem + sizeof(unsigned long long) * (size_t)n);
    } else {
        dist = reinterpret_cast<unsigned long long*>(orow);
        fl = reinterpret_cast<unsigned*>(smem);
    }
    unsigned* cur = fl;
    unsigned* nxt = fl + words;
    unsigned short* list = reinterpret_cast<unsigned short*>(fl + 2 * words);
    const unsigned long long INF_BITS = 0x7ff0000000000000ULL;
    for (int64_t v = tid; v < n; v += SSSP_THREADS) {
        const unsigned long long init = v == src ? 0ULL : INF_BITS;
        if (LDS_DIST) dist[v] = init;
        else __hip_atomic_store(&dist[v], init, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int w = tid; w < words; w += SSSP_THREADS) {
        cur[w] = (src >> 5) == w ? 1u << (src & 31) : 0u;
        nxt[w] = 0u;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int grp = tid >> 4, gl = tid & 15;          // 16 groups of 16 lanes
    for (int64_t round = 0; round <= n; ++round) {
        for (int w = tid; w < words; w += SSSP_THREADS) {
            unsigned m = cur[w];
            if (!m) continue;
            cur[w] = 0u;
            const int base = atomicAdd(&s_cnt, __popc(m));
            int k = 0;
            while (m) {
                list[base + k++] = (unsigned short)(w * 32 + __builtin_ctz(m));
                m &= m - 1u;
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        if (cnt == 0) break;
        for (int b = grp; b < cnt; b += SSSP_THREADS / 16) {
            const int v = list[b];
            const unsigned long long dvb = LDS_DIST ? dist[v]
                                                    : __hip_atomic_load(&dist[v], __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT);
            const double dv = __longlong_as_double((long long)dvb);
            const int e1 = indptr[v + 1];
            for (int e = indptr[v] + gl; e < e1; e += 16) {
                const int u = adj_v[e];
                const double nd = dv + adj_w[e];
                const unsigned long long nb = (unsigned long long)__double_as_longlong(nd);
                const unsigned long long old = atomicMin(&dist[u], nb);
                if (nb < old) atomicOr(&nxt[u >> 5], 1u << (u & 31));
            }
        }
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        unsigned* t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
    if (LDS_DIST) {
        for (int64_t v = tid; v < n; v += SSSP_THREADS)
            orow[v] = __longlong_as_double((long long)dist[v]);
    }
}

extern "C" int64_t al3d_apsp_workspace_bytes(int64_t n, int kq)
{
    // deg[n] + indptr[n+1] + cursor[n] (int) + adj_v[2 n kq] (int) + adj_w[2 n kq] (f64)
    int64_t ints = al3d_align((3 * n + 1) * 4, 256) + al3d_align(2 * n * kq * 4, 256);
    return ints + al3d_align(2 * n * kq * 8, 256);
}

extern "C" int al3d_apsp_knn_rows_f64(const double* knn_d, const int64_t* knn_i, int64_t n, int kq,
                                      int64_t row0, int64_t nrows, double* out, void* workspace,
                                      void* stream)
{
    AL3D_REQUIRE(knn_d && knn_i && out && workspace, "al3d_apsp_knn_f64: null pointer");
    AL3D_REQUIRE(kq >= 1 && kq <= 32, "al3d_apsp_knn_f64: kq must be in [1,32] (got %d)", kq);
    AL3D_REQUIRE(n >= 0 && n < 65536 && n * (int64_t)kq < (1LL << 30), "al3d_apsp_knn_f64: n must be < 65536");
    AL3D_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= n, "al3d_apsp_knn_f64: bad row range");
    if (n == 0 || nrows == 0) return AL3D_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    int* deg = reinterpret_cast<int*>(w);
    int* indptr = deg + n;
    int* cursor = indptr + n + 1;
    int* adj_v = reinterpret_cast<int*>(w + al3d_align((3 * n + 1) * 4, 256));
    double* adj_w = reinterpret_cast<double*>(w + al3d_align((3 * n + 1) * 4, 256) +
                                              al3d_align(2 * n * kq * 4, 256));
    if (hipMemsetAsync(deg, 0, sizeof(int) * (size_t)n, s) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "al3d_apsp_knn_f64: memset failed");
    const unsigned eb = (unsigned)al3d_cdiv(n * kq, 256);
    hipLaunchKernelGGL(adj_count_kernel, dim3(eb), dim3(256), 0, s, knn_d, knn_i, n, kq, deg);
    hipLaunchKernelGGL(adj_scan_kernel, dim3(1), dim3(1024), 0, s, deg, n, indptr, cursor);
    hipLaunchKernelGGL(adj_fill_kernel, dim3(eb), dim3(256), 0, s, knn_d, knn_i, n, kq, cursor, adj_v,
                       adj_w);
    // flag words (two bit sets) + the 16-bit node list, 16-byte aligned after the labels
    const size_t words = (size_t)((n + 31) / 32);
    const size_t lds_flags = al3d_align(words * 8 + (size_t)n * 2, 16), lds_full = (size_t)n * 8 + lds_flags;
    const size_t lds_cap = 158 * 1024;
    if (lds_full <= lds_cap) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sssp_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full);
        hipLaunchKernelGGL(sssp_kernel<true>, dim3((unsigned)nrows), dim3(SSSP_THREADS), lds_full, s,
                           indptr, adj_v, adj_w, n, row0, out);
    } else {
        AL3D_REQUIRE(lds_flags <= lds_cap, "al3d_apsp_knn_f64: n=%lld too large", (long long)n);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sssp_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_flags);
        hipLaunchKernelGGL(sssp_kernel<false>, dim3((unsigned)nrows), dim3(SSSP_THREADS), lds_flags, s,
                           indptr, adj_v, adj_w, n, row0, out);
    }
    AL3D_CHECK_LAUNCH("sssp_kernel");
    return AL3D_OK;
}

extern "C" int al3d_apsp_knn_f64(const double* knn_d, const int64_t* knn_i, int64_t n, int kq,
                                 double* out, void* workspace, void* stream)
{
    return al3d_apsp_knn_rows_f64(knn_d, knn_i, n, kq, 0, n, out, workspace, stream);
}
