// Epilogues of the uncertainty selectors: per-frame mean binary entropy of
// the post-NMS scores, entropy weighting of embeddings, min-max normalisation and the
// descending argsort that replaces the greedy loop in EntropySelector.
//
// Reference: det3d/selectors/entropy_selector.py:50-86,121-147, badge_selector.py:50-90,
// uwe_selector.py:51-111.
#include "al3d_common.h"

// H = -s log s - (1-s) log(1-s), mean over the kept boxes of the frame; empty frame -> NaN
// (torch.mean of an empty tensor).
__global__ __launch_bounds__(64) void frame_entropy_kernel(const float* __restrict__ scores,
                                                           const int* __restrict__ counts, int nt,
                                                           int post, float* __restrict__ out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    float sum = 0.f;
    int total = 0;
    for (int t = 0; t < nt; ++t) {
        const int c = counts[b * nt + t];
        total += c;
        for (int i = lane; i < c; i += 64) {
            const float s = scores[((int64_t)b * nt + t) * post + i];
            sum += -s * logf(s) - (1.0f - s) * logf(1.0f - s);
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) out[b] = sum / (float)total;      // 0/0 = NaN for an empty frame
}

extern "C" int al3d_frame_entropy_f32(const float* scores, const int* counts, int B, int nt, int post,
                                      float* out, void* stream)
{
    AL3D_REQUIRE(scores && counts && out && B >= 0 && nt >= 1 && post >= 1, "al3d_frame_entropy_f32: bad arguments");
    if (B == 0) return AL3D_OK;
    hipLaunchKernelGGL(frame_entropy_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, scores,
                       counts, nt, post, out);
    AL3D_CHECK_LAUNCH("frame_entropy_kernel");
    return AL3D_OK;
}

// out[i][c] = feats[i][c] * w[widx ? widx[i] : i]
__global__ void scale_rows_kernel(const float* __restrict__ feats, const float* __restrict__ w,
                                  const int64_t* __restrict__ widx, int64_t n, int c,
                                  float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * c) return;
    const int64_t i = e / c;
    out[e] = feats[e] * w[widx ? widx[i] : i];
}

extern "C" int al3d_scale_rows_f32(const float* feats, const float* w, const int64_t* widx, int64_t n,
                                   int c, float* out, void* stream)
{
    AL3D_REQUIRE(n >= 0 && c >= 1, "al3d_scale_rows_f32: bad sizes");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(feats && w && out, "al3d_scale_rows_f32: null pointer");
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)al3d_cdiv(n * c, 256)), dim3(256), 0,
                       (hipStream_t)stream, feats, w, widx, n, c, out);
    AL3D_CHECK_LAUNCH("scale_rows_kernel");
    return AL3D_OK;
}

// (x - min) / (max - min) with torch's NaN-propagating min/max; one workgroup.
__global__ __launch_bounds__(1024) void minmax_norm_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ out)
{
    __shared__ float s_min[16], s_max[16];
    __shared__ int s_nan;
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        if (v != v) s_nan = 1;
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = mn; s_max[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = s_min[0]; mx = s_max[0];
    for (int w = 1; w < 16; ++w) { mn = fminf(mn, s_min[w]); mx = fmaxf(mx, s_max[w]); }
    if (s_nan) { mn = __builtin_nanf(""); mx = mn; }
    const float den = mx - mn;
    for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = (x[i] - mn) / den;
}

extern "C" int al3d_minmax_norm_f32(const float* x, int64_t n, float* out, void* stream)
{
    AL3D_REQUIRE(n >= 0, "al3d_minmax_norm_f32: bad size");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(x && out, "al3d_minmax_norm_f32: null pointer");
    hipLaunchKernelGGL(minmax_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    AL3D_CHECK_LAUNCH("minmax_norm_kernel");
    return AL3D_OK;
}

// argsort descending (torch.argsort(-x)): NaN last, equal values by ascending index.
// keys = (order-preserving bits of x) << 32 | ~index, bitonic sort by one workgroup.
__device__ __forceinline__ unsigned long long sort_key(float v, unsigned idx)
{
    unsigned u;
    if (v != v) u = 0u;                                    // NaN: smallest key -> last
    else {
        u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // total order, larger float -> larger key
        if (u == 0u) u = 1u;
    }
    return ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - idx);
}

__global__ __launch_bounds__(1024) void argsort_desc_kernel(const float* __restrict__ x, int64_t n, int64_t np2,
                                                            unsigned long long* __restrict__ keys,
                                                            int64_t* __restrict__ out)
{
    for (int64_t i = threadIdx.x; i < np2; i += 1024) keys[i] = i < n ? sort_key(x[i], (unsigned)i) : 0ull;
    __syncthreads();
    for (int64_t size = 2; size <= np2; size <<= 1)
        for (int64_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (int64_t i = threadIdx.x; i < np2; i += 1024) {
                const int64_t j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = keys[i], b = keys[j];
                    if (desc ? a < b : a > b) { keys[i] = b; keys[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = (int64_t)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
}

extern "C" int64_t al3d_argsort_workspace_bytes(int64_t n)
{
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p * 8;
}

extern "C" int al3d_argsort_desc_f32(const float* x, int64_t n, int64_t* out_idx, void* workspace, void* stream)
{
    AL3D_REQUIRE(n >= 0 && n < (1LL << 31), "al3d_argsort_desc_f32: bad size");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(x && out_idx && workspace, "al3d_argsort_desc_f32: null pointer");
    int64_t p = 1;
    while (p < n) p <<= 1;
    hipLaunchKernelGGL(argsort_desc_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, p,
                       (unsigned long long*)workspace, out_idx);
    AL3D_CHECK_LAUNCH("argsort_desc_kernel");
    return AL3D_OK;
}

// PPAL (det3d/selectors/ppal_selector.py:99-109): class-weighted entropy SUM per frame,
// weight looked up by the merged label id.  Empty frame -> 0 (sum of an empty tensor).
__global__ __launch_bounds__(64) void frame_weighted_entropy_kernel(const float* __restrict__ scores,
                                                                    const int* __restrict__ labels,
                                                                    const int* __restrict__ counts, int nt,
                                                                    int post, const float* __restrict__ cw,
                                                                    int ncls, float* __restrict__ out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    float sum = 0.f;
    for (int t = 0; t < nt; ++t) {
        const int c = counts[b * nt + t];
        for (int i = lane; i < c; i += 64) {
            const int64_t o = ((int64_t)b * nt + t) * post + i;
            const float s = scores[o];
            const int l = labels[o];
            const float w = (l >= 0 && l < ncls) ? cw[l] : 0.f;
            sum += (-s * logf(s) - (1.0f - s) * logf(1.0f - s)) * w;
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) out[b] = sum;
}

extern "C" int al3d_frame_weighted_entropy_f32(const float* scores, const int* labels, const int* counts,
                                               int B, int nt, int post, const float* class_weight,
                                               int ncls, float* out, void* stream)
{
    AL3D_REQUIRE(B >= 0 && nt >= 1 && post >= 1 && ncls >= 1, "al3d_frame_weighted_entropy_f32: bad sizes");
    if (B == 0) return AL3D_OK;
    AL3D_REQUIRE(scores && labels && counts && class_weight && out, "al3d_frame_weighted_entropy_f32: null pointer");
    hipLaunchKernelGGL(frame_weighted_entropy_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, scores,
                       labels, counts, nt, post, class_weight, ncls, out);
    AL3D_CHECK_LAUNCH("frame_weighted_entropy_kernel");
    return AL3D_OK;
}

// PPAL pool restriction (ppal_selector.py:194-196): rows and columns of frames outside the
// candidate pool become -inf so the greedy can never reach them.
__global__ void mask_map_kernel(float* __restrict__ D, int64_t n, const unsigned char* __restrict__ keep)
{
    const int64_t i = blockIdx.y;
    const bool ki = keep[i] != 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        if (!ki || !keep[j]) D[i * n + j] = -__builtin_inff();
}

extern "C" int al3d_mask_map_f32(float* D, int64_t n, const unsigned char* keep, void* stream)
{
    AL3D_REQUIRE(n >= 0 && n < (1LL << 31), "al3d_mask_map_f32: bad n");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(D && keep, "al3d_mask_map_f32: null pointer");
    unsigned gx = (unsigned)al3d_cdiv(n, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mask_map_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, D, n, keep);
    AL3D_CHECK_LAUNCH("mask_map_kernel");
    return AL3D_OK;
}
