// Sweep merge on device: the reference builds every frame's point cloud on the
// host from the key-frame .bin and nine sweep .bin files -- drop the ring column, remove the
// points close to the sensor (sweeps only, in the sweep's own frame), move the sweep into the key
// frame with a float64 4x4, append the time lag (det3d/datasets/pipelines/loading.py:17-63,
// 98-126).  Here the raw files are uploaded back to back and three small kernels do the rest:
// flag -> exclusive scan -> compact + transform, preserving file order then point order.
//
// Arithmetic: |x| < d && |y| < d on the raw float32 values; x' = ((T0*x + T1*y) + T2*z) + T3 in
// float64 without contraction, rounded once to float32 (numpy stores the float64 dot product back
// into the float32 array); time = float32(time_lag).
#include "al3d_common.h"
#include "al3d_scan.h"

__device__ __forceinline__ int sweep_file_of(const int64_t* __restrict__ off, int nfiles, int64_t i)
{
    int f = 0;
    while (f + 1 < nfiles && i >= off[f + 1]) ++f;
    return f;
}

__global__ void sweep_flag_kernel(const float* __restrict__ raw, const int64_t* __restrict__ off, int nfiles,
                                  int64_t total, float min_distance, int* __restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = sweep_file_of(off, nfiles, i);
    const float x = raw[5 * i], y = raw[5 * i + 1];
    const bool close = fabsf(x) < min_distance && fabsf(y) < min_distance;
    flags[i] = (f == 0 || !close) ? 1 : 0;
}

__global__ void sweep_emit_kernel(const float* __restrict__ raw, const int64_t* __restrict__ off, int nfiles,
                                  int64_t total, const double* __restrict__ xform,
                                  const unsigned char* __restrict__ has_xform,
                                  const double* __restrict__ time_lag, const int* __restrict__ flags,
                                  const int* __restrict__ pos, float* __restrict__ out, int* __restrict__ out_count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (i == total - 1) out_count[0] = pos[i] + flags[i];
    if (!flags[i]) return;
    const int f = sweep_file_of(off, nfiles, i);
    float x = raw[5 * i], y = raw[5 * i + 1], z = raw[5 * i + 2];
    const float w = raw[5 * i + 3];
    if (f > 0 && has_xform[f]) {
        const double* t = xform + 12 * f;
        const double xd = x, yd = y, zd = z;
        const float nx = (float)(((t[0] * xd + t[1] * yd) + t[2] * zd) + t[3]);
        const float ny = (float)(((t[4] * xd + t[5] * yd) + t[6] * zd) + t[7]);
        const float nz = (float)(((t[8] * xd + t[9] * yd) + t[10] * zd) + t[11]);
        x = nx; y = ny; z = nz;
    }
    float* o = out + 5 * (int64_t)pos[i];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
    o[4] = f == 0 ? 0.0f : (float)time_lag[f];
}

extern "C" int64_t al3d_merge_sweeps_workspace_bytes(int64_t total_rows)
{
    const int64_t n = total_rows > 0 ? total_rows : 1;
    return 2 * al3d_align(n * 4, 256) + al3d_scan_workspace_bytes(n);
}

extern "C" int al3d_merge_sweeps_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                     const double* xform, const unsigned char* has_xform,
                                     const double* time_lag, float min_distance, float* out, int* out_count,
                                     void* workspace, void* stream)
{
    AL3D_REQUIRE(nfiles >= 1 && total_rows >= 0 && total_rows < (1LL << 31), "al3d_merge_sweeps_f32: bad sizes");
    AL3D_REQUIRE(out_count, "al3d_merge_sweeps_f32: null out_count");
    hipStream_t s = (hipStream_t)stream;
    if (total_rows == 0) {
        if (hipMemsetAsync(out_count, 0, 4, s) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_merge_sweeps_f32: memset failed");
        return AL3D_OK;
    }
    AL3D_REQUIRE(raw && file_off && xform && has_xform && time_lag && out && workspace,
                 "al3d_merge_sweeps_f32: null pointer");
    int* flags = (int*)workspace;
    int* pos = (int*)((unsigned char*)workspace + al3d_align(total_rows * 4, 256));
    void* scan_ws = (unsigned char*)workspace + 2 * al3d_align(total_rows * 4, 256);
    const unsigned blocks = (unsigned)al3d_cdiv(total_rows, 256);
    hipLaunchKernelGGL(sweep_flag_kernel, dim3(blocks), dim3(256), 0, s, raw, file_off, nfiles, total_rows,
                       min_distance, flags);
    int rc = al3d_exclusive_scan_i32(flags, pos, total_rows, scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(sweep_emit_kernel, dim3(blocks), dim3(256), 0, s, raw, file_off, nfiles, total_rows, xform,
                       has_xform, time_lag, flags, pos, out, out_count);
    AL3D_CHECK_LAUNCH("merge_sweeps");
    return AL3D_OK;
}
