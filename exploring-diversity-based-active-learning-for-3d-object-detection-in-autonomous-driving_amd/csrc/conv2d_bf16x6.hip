// Dense 2-D convolution with fp32-faithful arithmetic on the bf16 matrix cores (gfx950).
//
// Every fp32 operand is split exactly into three bf16 pieces x = x1 + x2 + x3 (8+8+8
// significand bits); a product x*w is formed as the six partial products of total order <= 2
// (x1w1, x1w2, x2w1, x1w3, x2w2, x3w1), each exact in fp32, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The three dropped terms are below 2^-25 |x w| -- smaller than the
// rounding of the fp32 product itself -- so results agree with an fp32 FMA chain to fp32
// rounding noise (tests compare both kernels against an fp64 reference).  Six bf16 MFMAs cost
// 6/16 of the fp32-input MFMA they replace (MI355X: v_mfma_f32_32x32x2_f32 runs at 1/16 of the
// bf16 rate), i.e. up to 2.67x the fp32 matrix-core roof.
//
// Same role, layouts and tiling as conv2d_mfma.hip (NHWC f32 activations in HBM; 128 px x 128
// ch tile; 4 waves 2x2; double-buffered LDS); activations are split while they are staged
// (registers -> LDS), weights are pre-split once into [3][Cout][taps][Cin] bf16.
#include "al3d_common.h"

#define C6_BM 128
#define C6_BN 128
#define C6_BK 16
#define C6_LDB 48          // bytes per LDS row: 16 bf16 (32 B) + 16 B pad (conflict-free b128 reads)
#define C6_TH 8
#define C6_TW 16

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Conv6Params {
    const float* in;        // [B, H, W, Cin] f32
    const __bf16* wgt;      // [3][Cout][taps][Cin] bf16 planes (hi, mid, lo)
    const float* scale;
    const float* shift;
    float* out;
    int B, H, W, Cin, Cout, OH, OW, ldc, coff;
    int ksize, stride, pad, relu;
    int tiles_x, tiles_y;
    int64_t plane;          // elements per weight plane = Cout * taps * Cin
};

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c)
{
    a = (__bf16)x;
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_bf16x6_kernel(Conv6Params p)
{
    // [buf][A|B][plane][row * 48 B]
    // (Loading the pre-split weight fragments straight from L2 into registers instead of staging
    // them was measured 17 % slower: fragment-shaped loads touch 32 lines for 1 KiB.)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][3][C6_BM * C6_LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int tile = blockIdx.x;
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = blockIdx.y * C6_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;
    const int wtaps = MODE == 0 ? taps : 4;

    // A staging: 128 rows x 16 ch f32 = 4 float4 per row -> 512 float4, 2 per thread
    const int aq = tid & 3, ar = tid >> 2;            // piece, row (0..63), +64 on pass 1
    int py[2], px[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = ar + 64 * i;
        py[i] = ty_ * C6_TH + m / C6_TW;
        px[i] = tx_ * C6_TW + m % C6_TW;
    }
    // B staging: per plane 128 rows x 16 bf16 = 2 x 16 B per row -> 256 pieces, 1 per thread
    const int bq = tid & 1, br = tid >> 1;
    const int kchunks = p.Cin / C6_BK;
    const int nsteps = taps * kchunks;

    float4 ra[2];
    uint4 rb[3];
    auto load_step = [&](int step) {
        const int tap = step / kchunks, c0 = (step - tap * kchunks) * C6_BK;
        const int ky = MODE == 0 ? tap / p.ksize : 0, kx = MODE == 0 ? tap % p.ksize : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int iy, ix;
            if (MODE == 0) { iy = py[i] * p.stride - p.pad + ky; ix = px[i] * p.stride - p.pad + kx; }
            else { iy = py[i]; ix = px[i]; }
            const bool ok = py[i] < MH && px[i] < MW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            ra[i] = ok ? *reinterpret_cast<const float4*>(
                             p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + c0 + 4 * aq)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int n = n0 + br;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            rb[pl] = n < p.Cout ? *reinterpret_cast<const uint4*>(
                                      p.wgt + pl * p.plane + ((int64_t)n * wtaps + tap0 + tap) * p.Cin + c0 + 8 * bq)
                                : make_uint4(0u, 0u, 0u, 0u);
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 a, bb, c; split3(v[e], a, bb, c); h[e] = a; m[e] = bb; l[e] = c; }
            const int off = (ar + 64 * i) * C6_LDB + 8 * aq;
            *reinterpret_cast<bf16x4*>(&lds[buf][0][0][off]) = h;
            *reinterpret_cast<bf16x4*>(&lds[buf][0][1][off]) = m;
            *reinterpret_cast<bf16x4*>(&lds[buf][0][2][off]) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(&lds[buf][1][pl][br * C6_LDB + 16 * bq]) = rb[pl];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_step(0);
    store_step(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_step(step + 1);
        bf16x8 a[3][2], bb[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[pl][t] = *reinterpret_cast<const bf16x8*>(&lds[buf][0][pl][(wm * 64 + t * 32 + fr) * C6_LDB + 16 * fh]);
                bb[pl][t] = *reinterpret_cast<const bf16x8*>(&lds[buf][1][pl][(wn * 64 + t * 32 + fr) * C6_LDB + 16 * fh]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest partial products first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], bb[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], bb[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], bb[2][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], bb[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], bb[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], bb[0][j], acc[i][j], 0, 0, 0);
            }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int y = ty_ * C6_TH + m / C6_TW, x = tx_ * C6_TW + m % C6_TW;
                if (y >= MH || x >= MW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v > 0.f ? v : 0.f;
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

// one-off weight split: f32 [count] -> bf16 [3][count]
__global__ void split_weights_kernel(const float* __restrict__ w, int64_t count, __bf16* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    __bf16 a, b, c;
    split3(w[i], a, b, c);
    out[i] = a; out[count + i] = b; out[2 * count + i] = c;
}

extern "C" int al3d_split_bf16x3(const float* w, int64_t count, void* out_bf16x3, void* stream)
{
    AL3D_REQUIRE(w && out_bf16x3 && count >= 0, "al3d_split_bf16x3: bad arguments");
    if (count == 0) return AL3D_OK;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, count, (__bf16*)out_bf16x3);
    AL3D_CHECK_LAUNCH("split_weights_kernel");
    return AL3D_OK;
}

static int conv6_check(const Conv6Params& p, const char* name)
{
    AL3D_REQUIRE(p.in && p.wgt && p.out, "%s: null pointer", name);
    AL3D_REQUIRE(p.B >= 1 && p.H >= 1 && p.W >= 1 && p.Cin >= 1 && p.Cout >= 1, "%s: bad shape", name);
    AL3D_REQUIRE(p.Cin % C6_BK == 0, "%s: Cin=%d must be a multiple of %d", name, p.Cin, C6_BK);
    AL3D_REQUIRE(p.coff >= 0 && p.coff + p.Cout <= p.ldc, "%s: channel window [%d,%d) exceeds ldc=%d",
                 name, p.coff, p.coff + p.Cout, p.ldc);
    AL3D_REQUIRE(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0,
                 "%s: in/wgt must be 16-byte aligned", name);
    return AL3D_OK;
}

extern "C" int al3d_conv2d_nhwc_bf16x6(const float* in, const void* wgt_bf16x3, const float* scale,
                                       const float* shift, float* out, int B, int H, int W, int Cin,
                                       int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                       int relu, void* stream)
{
    Conv6Params p;
    p.in = in; p.wgt = (const __bf16*)wgt_bf16x3; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ldc = ldc; p.coff = coff; p.relu = relu;
    AL3D_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0, "al3d_conv2d_nhwc_bf16x6: bad geometry");
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    AL3D_REQUIRE(p.OH >= 1 && p.OW >= 1, "al3d_conv2d_nhwc_bf16x6: empty output");
    p.plane = (int64_t)Cout * ksize * ksize * Cin;
    int rc = conv6_check(p, "al3d_conv2d_nhwc_bf16x6");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(p.OW, C6_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, C6_TH);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)al3d_cdiv(Cout, C6_BN), 1);
    hipLaunchKernelGGL(conv2d_bf16x6_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_bf16x6_kernel<conv>");
    return AL3D_OK;
}

extern "C" int al3d_deconv2x2_nhwc_bf16x6(const float* in, const void* wgt_bf16x3, const float* scale,
                                          const float* shift, float* out, int B, int H, int W, int Cin,
                                          int Cout, int ldc, int coff, int relu, void* stream)
{
    Conv6Params p;
    p.in = in; p.wgt = (const __bf16*)wgt_bf16x3; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 2; p.stride = 2; p.pad = 0; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = 2 * H; p.OW = 2 * W;
    p.plane = (int64_t)Cout * 4 * Cin;
    int rc = conv6_check(p, "al3d_deconv2x2_nhwc_bf16x6");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(W, C6_TW);
    p.tiles_y = (int)al3d_cdiv(H, C6_TH);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)al3d_cdiv(Cout, C6_BN), 4);
    hipLaunchKernelGGL(conv2d_bf16x6_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_bf16x6_kernel<deconv>");
    return AL3D_OK;
}
