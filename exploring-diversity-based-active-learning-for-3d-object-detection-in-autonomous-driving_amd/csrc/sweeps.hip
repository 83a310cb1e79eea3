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


// ---------------------------------------------------------------------------------------------
// Batched form for the streaming file loader (reader.cpp): the files of B frames back to back, one
// launch set per batch instead of per frame.  is_key[f] marks the key-frame file of a frame (no
// remove_close, time 0); frame_first_file[b] is the index of frame b's first file ([B+1] entries).
// out_frame_off[b] = first output point of frame b ([B+1] int64, computed here from the scan).
// optional range filter of the merged cloud (BEVFusion's PointsRangeFilter, transforms_3d.py:503-525 on
// base_points.py:208-232): a point stays when lo < p < hi STRICTLY on x, y, z of the TRANSFORMED float32 point
struct SweepRange { float lo[3], hi[3]; int on; };

// sweep -> key frame under rule 0 / 1 (see al3d_merge_sweeps_batch_rule_f32)
__device__ __forceinline__ void sweep_apply_xform(const double* __restrict__ t, int rule, float& x, float& y, float& z)
{
    const double xd = x, yd = y, zd = z;
    if (rule == 0) {
        const float nx = (float)(((t[0] * xd + t[1] * yd) + t[2] * zd) + t[3]);
        const float ny = (float)(((t[4] * xd + t[5] * yd) + t[6] * zd) + t[7]);
        const float nz = (float)(((t[8] * xd + t[9] * yd) + t[10] * zd) + t[11]);
        x = nx; y = ny; z = nz;
    } else {
        // BEVFusion (bevfusion/mmdet3d/datasets/pipelines/loading.py:222-226): `p[:, :3] = p[:, :3] @ R.T` stores the
        // float64 product back into the float32 array, then `p[:, :3] += t` adds the float64 translation to the
        // ROUNDED value and rounds again
        const float rx = (float)((t[0] * xd + t[1] * yd) + t[2] * zd);
        const float ry = (float)((t[4] * xd + t[5] * yd) + t[6] * zd);
        const float rz = (float)((t[8] * xd + t[9] * yd) + t[10] * zd);
        x = (float)((double)rx + t[3]); y = (float)((double)ry + t[7]); z = (float)((double)rz + t[11]);
    }
}

__device__ __forceinline__ int sweep_file_of_bs(const int64_t* __restrict__ off, int nfiles, int64_t i)
{
    int lo = 0, hi = nfiles - 1;                  // largest f with off[f] <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void sweep_flag_batch_kernel(const float* __restrict__ raw, const int64_t* __restrict__ off, int nfiles,
                                        int64_t total, const unsigned char* __restrict__ is_key, float min_distance,
                                        const double* __restrict__ xform, const unsigned char* __restrict__ has_xform,
                                        int rule, SweepRange rg, int* __restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = sweep_file_of_bs(off, nfiles, i);
    float x = raw[5 * i], y = raw[5 * i + 1];
    const bool close = fabsf(x) < min_distance && fabsf(y) < min_distance;
    bool keep = is_key[f] || !close;
    if (keep && rg.on) {
        float z = raw[5 * i + 2];
        if (!is_key[f] && has_xform[f]) sweep_apply_xform(xform + 12 * f, rule, x, y, z);
        keep = x > rg.lo[0] && y > rg.lo[1] && z > rg.lo[2] && x < rg.hi[0] && y < rg.hi[1] && z < rg.hi[2];
    }
    flags[i] = keep ? 1 : 0;
}

__global__ void sweep_emit_batch_kernel(const float* __restrict__ raw, const int64_t* __restrict__ off, int nfiles,
                                        int64_t total, const double* __restrict__ xform,
                                        const unsigned char* __restrict__ has_xform,
                                        const double* __restrict__ time_lag, const unsigned char* __restrict__ is_key,
                                        const int* __restrict__ flags, const int* __restrict__ pos,
                                        float* __restrict__ out, int rule)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || !flags[i]) return;
    const int f = sweep_file_of_bs(off, nfiles, i);
    float x = raw[5 * i], y = raw[5 * i + 1], z = raw[5 * i + 2];
    const float w = raw[5 * i + 3];
    const bool key = is_key[f] != 0;
    if (!key && has_xform[f]) sweep_apply_xform(xform + 12 * f, rule, x, y, z);
    float* o = out + 5 * (int64_t)pos[i];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
    o[4] = key ? 0.0f : (float)time_lag[f];
}

__global__ void sweep_frame_off_kernel(const int64_t* __restrict__ off, const int* __restrict__ frame_first_file, int B,
                                       int64_t total, const int* __restrict__ flags, const int* __restrict__ pos,
                                       int64_t* __restrict__ out_frame_off)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    const int64_t end = (int64_t)pos[total - 1] + flags[total - 1];
    const int64_t r = b < B ? off[frame_first_file[b]] : total;
    out_frame_off[b] = r < total ? (int64_t)pos[r] : end;
}

extern "C" int al3d_merge_sweeps_batch_range_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                                 const double* xform, const unsigned char* has_xform,
                                                 const double* time_lag, const unsigned char* is_key,
                                                 const int* frame_first_file, int n_frames, float min_distance, int rule,
                                                 const float* point_range, float* out, int64_t* out_frame_off,
                                                 void* workspace, void* stream);

extern "C" int al3d_merge_sweeps_batch_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                           const double* xform, const unsigned char* has_xform,
                                           const double* time_lag, const unsigned char* is_key,
                                           const int* frame_first_file, int n_frames, float min_distance,
                                           float* out, int64_t* out_frame_off, void* workspace, void* stream)
{
    return al3d_merge_sweeps_batch_range_f32(raw, file_off, nfiles, total_rows, xform, has_xform, time_lag, is_key,
                                             frame_first_file, n_frames, min_distance, 0, nullptr, out, out_frame_off, workspace,
                                             stream);
}

// rule 0: det3d's loader (one float64 4x4 product, rounded once); rule 1: BEVFusion's LoadPointsFromMultiSweeps (float64
// rotation rounded to float32, then the float64 translation added and rounded again); is_key files: no remove_close, time 0
// (BEVFusion's padded key-frame copies of an empty sweep list are ordinary files with has_xform 0 and time lag 0)
extern "C" int al3d_merge_sweeps_batch_rule_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                                const double* xform, const unsigned char* has_xform,
                                                const double* time_lag, const unsigned char* is_key,
                                                const int* frame_first_file, int n_frames, float min_distance, int rule,
                                                float* out, int64_t* out_frame_off, void* workspace, void* stream)
{
    return al3d_merge_sweeps_batch_range_f32(raw, file_off, nfiles, total_rows, xform, has_xform, time_lag, is_key,
                                             frame_first_file, n_frames, min_distance, rule, nullptr, out, out_frame_off,
                                             workspace, stream);
}

// ... followed by the pipeline's PointsRangeFilter (bevfusion/configs/nuscenes/default.yaml:233-235): point_range = HOST
// float[6] (x_min, y_min, z_min, x_max, y_max, z_max) or NULL (no filter); a merged point stays when it lies STRICTLY inside
extern "C" int al3d_merge_sweeps_batch_range_f32(const float* raw, const int64_t* file_off, int nfiles, int64_t total_rows,
                                                 const double* xform, const unsigned char* has_xform,
                                                 const double* time_lag, const unsigned char* is_key,
                                                 const int* frame_first_file, int n_frames, float min_distance, int rule,
                                                 const float* point_range, float* out, int64_t* out_frame_off,
                                                 void* workspace, void* stream)
{
    SweepRange rg = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0};
    if (point_range) {
        for (int d = 0; d < 3; ++d) { rg.lo[d] = point_range[d]; rg.hi[d] = point_range[3 + d]; }
        rg.on = 1;
    }
    AL3D_REQUIRE(rule == 0 || rule == 1, "al3d_merge_sweeps_batch_rule_f32: rule 0 (det3d) or 1 (BEVFusion)");
    AL3D_REQUIRE(nfiles >= 1 && n_frames >= 1 && total_rows >= 0 && total_rows < (1LL << 31),
                 "al3d_merge_sweeps_batch_f32: bad sizes");
    AL3D_REQUIRE(out_frame_off, "al3d_merge_sweeps_batch_f32: null out_frame_off");
    hipStream_t s = (hipStream_t)stream;
    if (total_rows == 0) {
        if (hipMemsetAsync(out_frame_off, 0, 8 * (size_t)(n_frames + 1), s) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_merge_sweeps_batch_f32: memset failed");
        return AL3D_OK;
    }
    AL3D_REQUIRE(raw && file_off && xform && has_xform && time_lag && is_key && frame_first_file && out && workspace,
                 "al3d_merge_sweeps_batch_f32: null pointer");
    int* flags = (int*)workspace;
    int* pos = (int*)((unsigned char*)workspace + al3d_align(total_rows * 4, 256));
    void* scan_ws = (unsigned char*)workspace + 2 * al3d_align(total_rows * 4, 256);
    const unsigned blocks = (unsigned)al3d_cdiv(total_rows, 256);
    hipLaunchKernelGGL(sweep_flag_batch_kernel, dim3(blocks), dim3(256), 0, s, raw, file_off, nfiles, total_rows,
                       is_key, min_distance, xform, has_xform, rule, rg, flags);
    int rc = al3d_exclusive_scan_i32(flags, pos, total_rows, scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(sweep_emit_batch_kernel, dim3(blocks), dim3(256), 0, s, raw, file_off, nfiles, total_rows,
                       xform, has_xform, time_lag, is_key, flags, pos, out, rule);
    hipLaunchKernelGGL(sweep_frame_off_kernel, dim3((unsigned)al3d_cdiv(n_frames + 1, 256)), dim3(256), 0, s, file_off,
                       frame_first_file, n_frames, total_rows, flags, pos, out_frame_off);
    AL3D_CHECK_LAUNCH("merge_sweeps_batch");
    return AL3D_OK;
}


// ---------------------------------------------------------------------------------------------
// Runtime helper: a HIP stream whose kernels may only run on `n_cus` of the device's compute units.
// The sweep's side stream (next batch's voxelizer + rulebook: small latency-bound kernels with random
// HBM traffic) and the decode/NMS stream use it so that their work keeps a bounded number of memory
// requests in flight next to the main stream's convolutions.  The stream is created once per process and
// lives until exit (torch wraps it as an ExternalStream).
extern "C" int al3d_stream_create_cu_mask(int n_cus, int first_cu, void** out_stream)
{
    AL3D_REQUIRE(out_stream, "al3d_stream_create_cu_mask: null pointer");
    int dev = 0, total = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "al3d_stream_create_cu_mask: cannot query the device");
    AL3D_REQUIRE(n_cus >= 1 && first_cu >= 0 && first_cu + n_cus <= total,
                 "al3d_stream_create_cu_mask: %d CUs from %d do not fit the device's %d", n_cus, first_cu, total);
    uint32_t mask[16] = {0};
    AL3D_REQUIRE(total <= 512, "al3d_stream_create_cu_mask: more than 512 compute units");
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    if (e != hipSuccess) return al3d_fail(AL3D_ELAUNCH, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    *out_stream = (void*)s;
    return AL3D_OK;
}
