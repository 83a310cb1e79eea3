// Dev probe: f16 MFMA throughput on random operands, 32x32x16 vs 16x16x32 (same FLOPs per wave-loop).
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const f16x8* __restrict__ src, float* __restrict__ out, int iters)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) % 65536]; b[i] = src[(tid * 8 + 4 + i) % 65536]; }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[16] = {};                 // same accumulator footprint; 16x16x32 has half the flops per instruction
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + j) & 3], b[(i >> 2)], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[tid] = s;
}

int main(int argc, char** argv)
{
    const int zero = argc > 1 && atoi(argv[1]) == 0 ? 1 : 0;
    std::vector<_Float16> h(65536 * 8);
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 997.0f);
    f16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4 * 256 * 1024 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int blocks = 1024, iters = 20000;
    for (int shape : {32, 16, 32, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // flops: 32-shape: 16 mfma x 32*32*16*2 per iter per wave; 16-shape: 32 mfma x 16*16*32*2
        const double fl = (double)blocks * 4 * iters * (shape == 32 ? 16.0 * 32768 : 32.0 * 16384);
        printf("%s data, %dx%dx%d: %.1f ms  %.0f TFLOP/s\n", zero ? "zero" : "random", shape, shape, shape == 32 ? 16 : 32, ms, fl / ms / 1e9);
    }
    return 0;
}
