// Dev probe (GPU box): how does rocprofv3's FETCH_SIZE count the load shapes this repo's kernels use?
// Every kernel reads each byte of a 2 GiB buffer (8x the 256 MiB Infinity Cache) exactly once, so the counted / actual
// ratio is the correction factor for that shape (MI355X_MICROARCH.md: 16 B/lane coalesced streaming loads and LDS-DMA read
// 1/2; other shapes "uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/probes/fetch_calib
// Shapes (S = contiguous bytes a lane group fetches together, groups placed pseudo-randomly unless "stream"):
//   stream16 / stream4   fully coalesced 16 B / 4 B per lane
//   seg<S>               S/16 adjacent lanes read one S-byte segment with dwordx4; segments permuted (64 B: the dense
//                        kernels' halo pieces; 128 / 256 / 512 B: whole rows of 32 / 64 / 128 channels)
//   frag32               lane (r, h) reads 32 B of its own 512-byte row per step (sp_conv_wave2's fragment-shaped gather)
//   dma<0> / dma<64> / dma<128> / dma<256>   global_load_lds_dwordx4 (LDS-DMA): coalesced 1 KiB pieces; 4 / 8 / 16 lanes per segment
//   walk64<1> / <0>      the dense kernels' K loop: 64-byte pieces of 2 KiB pixel records, chunk after chunk (LDS-DMA / registers)
//   x3_sector            one 12-byte dwordx3 probe per 64-byte sector (sp_table_rows27's lookups): 12 of 64 bytes used
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void sink(float4 v, float* out)
{
    if (v.x == 1.2345e-30f && v.y == -7.1e-31f) out[0] = v.z + v.w;          // never true on zero-filled memory
}

__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ buf, int64_t n16, float* out)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const float4 v = buf[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    sink(acc, out);
}

__global__ __launch_bounds__(256) void stream4(const float* __restrict__ buf, int64_t n4, float* out)
{
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) acc += buf[i];
    if (acc == 1.2345e-30f) out[0] = acc;                     // never true on zero-filled memory
}

// S-byte segments at permuted positions: thread t handles 16-byte piece (t % LPS) of segment perm(t / LPS)
template <int S>
__global__ __launch_bounds__(256) void seg(const float4* __restrict__ buf, int64_t nseg, int64_t mult, float* out)
{
    constexpr int LPS = S / 16;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nseg * LPS; t += (int64_t)gridDim.x * 256) {
        const int64_t s = ((t / LPS) * mult) & (nseg - 1);
        const float4 v = buf[s * LPS + (t % LPS)];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    sink(acc, out);
}

// fragment-shaped: a wave owns 32 permuted 512-byte rows; step k: lane (r, h) reads bytes [64 k + 32 h, +32) of row r
__global__ __launch_bounds__(256) void frag32(const float4* __restrict__ buf, int64_t nrows, int64_t mult, float* out)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    float4 acc = make_float4(0, 0, 0, 0);
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t t = w0; t < nrows / 32; t += nw) {
        const int64_t row = ((t * 32 + r) * mult) & (nrows - 1);
        const float4* p = buf + row * 32 + 2 * h;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 a = p[4 * k], b = p[4 * k + 1];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        }
    }
    sink(acc, out);
}

// LDS-DMA: each wave moves 1 KiB pieces into its own LDS slot; S = 0: coalesced stream, else S-byte segments permuted
template <int S>
__global__ __launch_bounds__(256) void dma(const char* __restrict__ buf, int64_t bytes, int64_t mult, float* out)
{
    __shared__ __attribute__((aligned(1024))) char slot[4][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t npieces = bytes / 1024, nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
    constexpr int LPS = S ? S / 16 : 64;                      // lanes per segment
    const int64_t nseg = S ? bytes / S : 0;
    int q = 0;
    for (int64_t p = w0; p < npieces; p += nw) {
        const char* src;
        if (S == 0) src = buf + p * 1024 + lane * 16;
        else {
            const int64_t s = ((p * (64 / LPS) + lane / LPS) * mult) & (nseg - 1);
            src = buf + s * S + (lane % LPS) * 16;
        }
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(slot[wave] + (q & 3) * 1024), 16, 0, 0);
        ++q;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (slot[wave][lane] == 77) out[0] = 1.f;
}

// The dense kernels' K loop: a wave owns 16 consecutive 2 KiB pixel records and walks them 64 bytes (one 16-channel chunk) per
// step, 4 lanes per piece, with the step time of a real launch between two chunks (s_sleep): both halves of every 128-byte
// line ARE fetched, a step apart.  DMA = 1: global_load_lds (conv2d_f16x3_dma2_kernel), 0: registers (the 3x3 halo loads).
template <int DMA>
__global__ __launch_bounds__(256) void walk64(const char* __restrict__ buf, int64_t bytes, float* out)
{
    __shared__ __attribute__((aligned(1024))) char slot[4][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ngroups = bytes / (16 * 2048), nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
    float4 acc = make_float4(0, 0, 0, 0);
    int q = 0;
    for (int64_t g = w0; g < ngroups; g += nw) {
        const char* row = buf + (g * 16 + (lane >> 2)) * 2048 + (lane & 3) * 16;
        for (int c = 0; c < 32; ++c) {
            if (DMA) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(row + c * 64), (lds_void*)(slot[wave] + (q & 3) * 1024), 16, 0, 0);
                ++q;
            } else {
                const float4 v = *reinterpret_cast<const float4*>(row + c * 64);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            __builtin_amdgcn_s_sleep(30);                     // ~2k cycles: a step of the real kernels
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DMA) { if (slot[wave][lane] == 77) out[0] = 1.f; }
    else sink(acc, out);
}

typedef int i32x3 __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(256) void x3_sector(const char* __restrict__ buf, int64_t nsect, int64_t mult, float* out)
{
    int acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nsect; t += (int64_t)gridDim.x * 256) {
        const int64_t s = (t * mult) & (nsect - 1);
        const i32x3 v = *reinterpret_cast<const i32x3*>(buf + s * 64 + 16);
        acc += v[0] + v[1] + v[2];
    }
    if (acc == 0x7fffffff) out[0] = 2.f;
}

int main()
{
    const int64_t bytes = 2ll << 30;
    char* buf;
    float* out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipMemset(out, 0, 256));
    const int grid = 256 * 8;
    const int64_t mult = 2654435761ll | 1;                     // odd: a bijection on any power-of-two range
    printf("buffer %lld bytes; every kernel reads each byte once (x3_sector: 12 of every 64)\n", (long long)bytes);
    for (int rep = 0; rep < 2; ++rep) {                        // rep 0 also warms clocks; both are counted
        hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(stream4, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(seg<32>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 32, mult, out);
        hipLaunchKernelGGL(seg<64>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 64, mult, out);
        hipLaunchKernelGGL(seg<128>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 128, mult, out);
        hipLaunchKernelGGL(seg<256>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 256, mult, out);
        hipLaunchKernelGGL(seg<512>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 512, mult, out);
        hipLaunchKernelGGL(frag32, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 512, mult, out);
        hipLaunchKernelGGL(dma<0>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, mult, out);
        hipLaunchKernelGGL(dma<64>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, mult, out);
        hipLaunchKernelGGL(dma<128>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, mult, out);
        hipLaunchKernelGGL(dma<256>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, mult, out);
        hipLaunchKernelGGL(walk64<1>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, out);
        hipLaunchKernelGGL(walk64<0>, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes, out);
        hipLaunchKernelGGL(x3_sector, dim3(grid), dim3(256), 0, 0, (const char*)buf, bytes / 64, mult, out);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipGetLastError());
    printf("done\n");
    return 0;
}
