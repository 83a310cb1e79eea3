// Camera images of a BEVFusion sample on device: the test branch of the reference's image pipeline
// (bevfusion/mmdet3d/datasets/pipelines/loading.py:19-83 LoadMultiViewImageFromFiles -> decoded RGB frames;
// transforms_3d.py:26-122 ImageAug3D: img.resize(resize_dims) -> img.crop(crop) [no flip, rotate(0)];
// transforms_3d.py:903-920 ImageNormalize: ToTensor + Normalize).
//
// `Image.resize` without a filter argument is PIL's BICUBIC convolution resize (the reference pins Pillow 8.4.0,
// bevfusion/README.md:71), an integer algorithm on 8-bit pixels (Pillow src/libImaging/Resample.c; Pillow itself is not in
// /root/reference -- restated from its published algorithm and pinned against the installed Pillow in tests):
//   * per output coordinate the filter window [xmin, xmin + n) and its n weights, computed in double, normalised to sum 1
//     and rounded to 22-bit fixed point (al3d_image_resample_coeffs, host);
//   * horizontal pass: out = clip8((2^21 + sum px * k) >> 22) per channel into an 8-bit intermediate image, for the source
//     rows the vertical pass needs; vertical pass: the same over rows.
// Only the pixels of the crop window are computed.  The second pass converts like ToTensor / Normalize:
// (float(u8) / 255 - mean) / std in float32 with IEEE division, and writes channels-last float32 (the layout the token
// kernels read).  Bit-identical to PIL's resize + crop and to the torch float32 expression.
#include "al3d_common.h"
#include <math.h>

#define IMG_PRECISION_BITS 22

static inline double img_bicubic(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static inline double img_bilinear(double x)
{
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

// taps per output coordinate for al3d_image_resample_coeffs (filter: 2 = bilinear, 3 = bicubic, PIL's constants)
extern "C" int al3d_image_resample_ksize(int in_size, int out_size, int filter)
{
    if (in_size <= 0 || out_size <= 0 || (filter != 2 && filter != 3)) return -1;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == 3 ? 2.0 : 1.0) * filterscale;
    return (int)ceil(support) * 2 + 1;
}

// bounds [out_size][2] = (first source index, tap count), coeffs [out_size][ksize] 22-bit fixed point (unused taps 0)
extern "C" int al3d_image_resample_coeffs(int in_size, int out_size, int filter, int* bounds, int* coeffs)
{
    const int ksize = al3d_image_resample_ksize(in_size, out_size, filter);
    AL3D_REQUIRE(ksize > 0 && bounds && coeffs, "al3d_image_resample_coeffs: bad arguments (filter 2 = bilinear, 3 = bicubic)");
    const double in0 = 0.0, in1 = (double)(float)in_size;
    double scale, filterscale;
    filterscale = scale = (in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == 3 ? 2.0 : 1.0) * filterscale;
    double* k = (double*)malloc(sizeof(double) * ksize);
    if (!k) return al3d_fail(AL3D_EINVAL, "al3d_image_resample_coeffs: out of memory");
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double arg = (x + xmin - center + 0.5) * ss;
            const double w = filter == 3 ? img_bicubic(arg) : img_bilinear(arg);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
        for (x = 0; x < ksize; ++x)
            coeffs[(int64_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << IMG_PRECISION_BITS))
                                                       : (int)(0.5 + k[x] * (1 << IMG_PRECISION_BITS));
    }
    free(k);
    return AL3D_OK;
}

__device__ __forceinline__ int img_clip8(int v)
{
    v >>= IMG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: temp[img][ty][tx][c], ty over source rows [row0, row0 + nrows), tx over resized columns [cx0, cx0 + fW)
__global__ __launch_bounds__(256) void img_resize_h_kernel(const unsigned char* __restrict__ src, int nimg, int H, int W,
                                                           int row0, int nrows, int cx0, int fW,
                                                           const int* __restrict__ hb, const int* __restrict__ hk, int ksize,
                                                           unsigned char* __restrict__ temp)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)nimg * nrows * fW) return;
    const int tx = (int)(t % fW), ty = (int)((t / fW) % nrows), im = (int)(t / ((int64_t)fW * nrows));
    const int xx = cx0 + tx;
    const int xmin = hb[2 * xx], n = hb[2 * xx + 1];
    const int* k = hk + (int64_t)xx * ksize;
    const unsigned char* p = src + (((int64_t)im * H + row0 + ty) * W + xmin) * 3;
    int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        s0 += p[3 * x] * kv;
        s1 += p[3 * x + 1] * kv;
        s2 += p[3 * x + 2] * kv;
    }
    unsigned char* o = temp + t * 3;
    o[0] = (unsigned char)img_clip8(s0);
    o[1] = (unsigned char)img_clip8(s1);
    o[2] = (unsigned char)img_clip8(s2);
}

// vertical pass + ToTensor / Normalize: out[img][oy][ox][c] f32, oy over resized rows [cy0, cy0 + fH)
__global__ __launch_bounds__(256) void img_resize_v_norm_kernel(const unsigned char* __restrict__ temp, int nimg, int row0,
                                                                int nrows, int cy0, int fH, int fW,
                                                                const int* __restrict__ vb, const int* __restrict__ vk,
                                                                int ksize, float m0, float m1, float m2, float d0, float d1,
                                                                float d2, float* __restrict__ out,
                                                                unsigned char* __restrict__ out_u8)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)nimg * fH * fW) return;
    const int ox = (int)(t % fW), oy = (int)((t / fW) % fH), im = (int)(t / ((int64_t)fW * fH));
    const int yy = cy0 + oy;
    const int ymin = vb[2 * yy] - row0, n = vb[2 * yy + 1];
    const int* k = vk + (int64_t)yy * ksize;
    const unsigned char* p = temp + (((int64_t)im * nrows + ymin) * fW + ox) * 3;
    int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int kv = k[y];
        const unsigned char* q = p + (int64_t)y * fW * 3;
        s0 += q[0] * kv;
        s1 += q[1] * kv;
        s2 += q[2] * kv;
    }
    const int c0 = img_clip8(s0), c1 = img_clip8(s1), c2 = img_clip8(s2);
    if (out_u8) {
        out_u8[t * 3] = (unsigned char)c0; out_u8[t * 3 + 1] = (unsigned char)c1; out_u8[t * 3 + 2] = (unsigned char)c2;
    }
    if (out) {
        // ToTensor: float32(u8) / 255; Normalize: (t - mean) / std -- float32, correctly rounded divisions
        out[t * 3] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c0, 255.0f), m0), d0);
        out[t * 3 + 1] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c1, 255.0f), m1), d1);
        out[t * 3 + 2] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c2, 255.0f), m2), d2);
    }
}

extern "C" int64_t al3d_image_aug_workspace_bytes(int nimg, int in_h, int out_w)
{
    return al3d_align((int64_t)(nimg > 0 ? nimg : 1) * in_h * out_w * 3, 256);
}

// imgs [nimg][H][W][3] u8 (decoded RGB) -> resize to (rH, rW) with the given tables (al3d_image_resample_coeffs of (W, rW) and
// (H, rH)), crop the window [crop_x, crop_x + fW) x [crop_y, crop_y + fH), normalise -> out [nimg][fH][fW][3] f32.
// out_u8 (optional, may be NULL): the cropped 8-bit image PIL would hold before ImageNormalize.
extern "C" int al3d_image_aug_normalize_u8(const unsigned char* imgs, int nimg, int H, int W, int rH, int rW, int crop_x,
                                           int crop_y, int fH, int fW, const int* h_bounds, const int* h_coeffs, int h_ksize,
                                           const int* v_bounds, const int* v_coeffs, int v_ksize, const float* mean3,
                                           const float* std3, int row_first, int row_count, void* workspace, float* out,
                                           unsigned char* out_u8, void* stream)
{
    AL3D_REQUIRE(nimg >= 0 && H > 0 && W > 0 && rH > 0 && rW > 0 && fH > 0 && fW > 0, "al3d_image_aug_normalize_u8: bad sizes");
    AL3D_REQUIRE(crop_x >= 0 && crop_y >= 0 && crop_x + fW <= rW && crop_y + fH <= rH,
                 "al3d_image_aug_normalize_u8: the crop window must lie inside the resized image (PIL pads otherwise: not built)");
    AL3D_REQUIRE(row_first >= 0 && row_count > 0 && row_first + row_count <= H,
                 "al3d_image_aug_normalize_u8: row_first / row_count = the source rows the crop's vertical taps cover");
    if (nimg == 0) return AL3D_OK;
    AL3D_REQUIRE(imgs && h_bounds && h_coeffs && v_bounds && v_coeffs && mean3 && std3 && workspace && (out || out_u8),
                 "al3d_image_aug_normalize_u8: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned char* temp = (unsigned char*)workspace;
    const int64_t n1 = (int64_t)nimg * row_count * fW, n2 = (int64_t)nimg * fH * fW;
    hipLaunchKernelGGL(img_resize_h_kernel, dim3((unsigned)al3d_cdiv(n1, 256)), dim3(256), 0, s, imgs, nimg, H, W, row_first,
                       row_count, crop_x, fW, h_bounds, h_coeffs, h_ksize, temp);
    hipLaunchKernelGGL(img_resize_v_norm_kernel, dim3((unsigned)al3d_cdiv(n2, 256)), dim3(256), 0, s, temp, nimg, row_first,
                       row_count, crop_y, fH, fW, v_bounds, v_coeffs, v_ksize, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2], out, out_u8);
    AL3D_CHECK_LAUNCH("al3d_image_aug_normalize_u8");
    return AL3D_OK;
}
