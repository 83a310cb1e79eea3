// Three-kernel exclusive scan (block sums -> scan of block sums -> downsweep),
// 2048 items per 1024-thread workgroup; the middle kernel is one workgroup that
// walks the block sums in 1024-wide chunks with a running base, so any n works.
#include "al3d_common.h"
#include "al3d_scan.h"

#define SC_THREADS 1024
#define SC_ITEMS 2048

__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave, int& total)
{
    // inclusive scan inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    if (wave == 0) {
        int w = lane < SC_THREADS / 64 ? s_wave[lane] : 0;
        for (int off = 1; off < SC_THREADS / 64; off <<= 1) {
            int y = __shfl_up(w, off);
            if (lane >= off) w += y;
        }
        if (lane < SC_THREADS / 64) s_wave[lane] = w;   // inclusive wave totals
    }
    __syncthreads();
    const int base = wave > 0 ? s_wave[wave - 1] : 0;
    total = s_wave[SC_THREADS / 64 - 1];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(SC_THREADS) void scan_block_sums(const int* __restrict__ in, int64_t n,
                                                              int* __restrict__ sums)
{
    __shared__ int s_wave[SC_THREADS / 64];
    const int64_t i0 = (int64_t)blockIdx.x * SC_ITEMS + 2 * threadIdx.x;
    int v = 0;
    if (i0 < n) v += in[i0];
    if (i0 + 1 < n) v += in[i0 + 1];
    int total;
    block_exclusive_scan(v, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SC_THREADS) void scan_sums(int* __restrict__ sums, int64_t nb)
{
    __shared__ int s_wave[SC_THREADS / 64];
    int base = 0;
    for (int64_t c0 = 0; c0 < nb; c0 += SC_THREADS) {
        const int64_t i = c0 + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, s_wave, total);
        if (i < nb) sums[i] = base + ex;
        base += total;
    }
}

__global__ __launch_bounds__(SC_THREADS) void scan_downsweep(const int* __restrict__ in, int64_t n,
                                                             const int* __restrict__ sums,
                                                             int* __restrict__ out)
{
    __shared__ int s_wave[SC_THREADS / 64];
    const int64_t i0 = (int64_t)blockIdx.x * SC_ITEMS + 2 * threadIdx.x;
    const int a = i0 < n ? in[i0] : 0;
    const int b = i0 + 1 < n ? in[i0 + 1] : 0;
    int total;
    const int ex = block_exclusive_scan(a + b, s_wave, total) + sums[blockIdx.x];
    if (i0 < n) out[i0] = ex;
    if (i0 + 1 < n) out[i0 + 1] = ex + a;
}

int64_t al3d_scan_workspace_bytes(int64_t n) { return al3d_align((al3d_cdiv(n, SC_ITEMS) + 1) * 4, 256); }

int al3d_exclusive_scan_i32(const int* in, int* out, int64_t n, void* ws, hipStream_t stream)
{
    if (n <= 0) return AL3D_OK;
    const int64_t nb = al3d_cdiv(n, SC_ITEMS);
    int* sums = (int*)ws;
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(SC_THREADS), 0, stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums, dim3(1), dim3(SC_THREADS), 0, stream, sums, nb);
    hipLaunchKernelGGL(scan_downsweep, dim3((unsigned)nb), dim3(SC_THREADS), 0, stream, in, n, sums, out);
    AL3D_CHECK_LAUNCH("exclusive_scan");
    return AL3D_OK;
}
