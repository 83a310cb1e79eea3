// Dense 2-D convolution for the SECOND-style neck and the anchor head as an
// implicit GEMM on the fp32-input matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate, 64 FLOP/clk/SIMD).
//
// Reference layers (det3d/models/necks/rpn.py:124-159, bbox_heads/mg_head.py:215-231):
//   ZeroPad2d(1)+Conv2d(3x3, stride 1|2) / Conv2d(3x3, pad 1) / Conv2d(1x1) /
//   ConvTranspose2d(2x2, stride 2), each followed by eval BatchNorm2d(eps 1e-3)+ReLU
//   (neck) or a bias (head).  BN is folded to per-channel scale/shift and applied in
//   the epilogue together with the ReLU, so activations make one HBM round trip.
//
// Layout: activations NHWC f32 (pixel-major, channels contiguous); weights
// [Cout][tap][Cin] so both GEMM operands are "row x contiguous-K".
// Tiling: 128 output pixels (8x16 patch) x 128 output channels per 256-thread
// workgroup; 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs); K-chunks of 32 channels per tap, register-staged into double-buffered
// LDS tiles padded to 36 floats per row (conflict-free ds_read_b128 fragments).
#include "al3d_common.h"

#define CV_BM 128
#define CV_BN 128
#define CV_BK 32
#define CV_LD (CV_BK + 4)
#define CV_TH 8
#define CV_TW 16

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in;      // [B, H, W, Cin]
    const float* wgt;     // [Cout, taps, Cin]
    const float* scale;   // [Cout] or null (=1)
    const float* shift;   // [Cout] or null (=0)
    float* out;           // [B, OH, OW, ldc] written at channel offset coff
    int B, H, W, Cin, Cout, OH, OW, ldc, coff;
    int ksize, stride, pad;  // conv mode
    int relu;
    int tiles_x, tiles_y;    // 8x16 patches per image (over the GEMM-M pixel grid)
};

// MODE 0: convolution (ksize x ksize, stride, pad).  MODE 1: ConvTranspose2d 2x2 stride 2
// (blockIdx.z = output phase dy*2+dx; GEMM-M runs over *input* pixels).
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_mfma_kernel(ConvParams p)
{
    __shared__ __attribute__((aligned(16))) float lds[2][2][CV_BM * CV_LD];  // [buf][A|B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // which pixel patch / image
    int tile = blockIdx.x;
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = blockIdx.y * CV_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;  // GEMM-M pixel grid
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;  // deconv: fixed weight tap per launch slice

    // staging role: 8 threads per row (8 x float4 = 32 floats), 32 rows per pass, 4 passes
    const int sq = tid & 7, sr = tid >> 3;
    int py[4], px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = sr + 32 * i;
        py[i] = ty_ * CV_TH + m / CV_TW;
        px[i] = tx_ * CV_TW + m % CV_TW;
    }
    const int kchunks = p.Cin / CV_BK;
    const int nsteps = taps * kchunks;

    float4 ra[4], rb[4];
    auto load_step = [&](int step) {
        const int tap = step / kchunks, c0 = (step - tap * kchunks) * CV_BK;
        const int ky = MODE == 0 ? tap / p.ksize : 0, kx = MODE == 0 ? tap % p.ksize : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int iy, ix;
            if (MODE == 0) { iy = py[i] * p.stride - p.pad + ky; ix = px[i] * p.stride - p.pad + kx; }
            else { iy = py[i]; ix = px[i]; }
            const bool ok = py[i] < MH && px[i] < MW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            ra[i] = ok ? *reinterpret_cast<const float4*>(
                             p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + c0 + 4 * sq)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + sr + 32 * i;
            rb[i] = n < p.Cout ? *reinterpret_cast<const float4*>(
                                     p.wgt + ((int64_t)n * (MODE == 0 ? taps : 4) + tap0 + tap) * p.Cin +
                                     c0 + 4 * sq)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = sr + 32 * i;
            *reinterpret_cast<float4*>(&lds[buf][0][row * CV_LD + 4 * sq]) = ra[i];
            *reinterpret_cast<float4*>(&lds[buf][1][row * CV_LD + 4 * sq]) = rb[i];
        }
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
        const float* As = &lds[buf][0][(wm * 64 + fr) * CV_LD + 4 * fh];
        const float* Bs = &lds[buf][1][(wn * 64 + fr) * CV_LD + 4 * fh];
#pragma unroll
        for (int kg = 0; kg < CV_BK / 8; ++kg) {
            const float4 a0 = *reinterpret_cast<const float4*>(As + kg * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * CV_LD + kg * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(Bs + kg * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * CV_LD + kg * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // epilogue: scale/shift (+ReLU), NHWC store.  C/D map: col = lane&31,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5).
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
                const int y = ty_ * CV_TH + m / CV_TW, x = tx_ * CV_TW + m % CV_TW;
                if (y >= MH || x >= MW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

static int conv_check(const ConvParams& p, const char* name)
{
    AL3D_REQUIRE(p.in && p.wgt && p.out, "%s: null pointer", name);
    AL3D_REQUIRE(p.B >= 1 && p.H >= 1 && p.W >= 1 && p.Cin >= 1 && p.Cout >= 1, "%s: bad shape", name);
    AL3D_REQUIRE(p.Cin % CV_BK == 0, "%s: Cin=%d must be a multiple of %d", name, p.Cin, CV_BK);
    AL3D_REQUIRE(p.coff >= 0 && p.coff + p.Cout <= p.ldc, "%s: channel window [%d,%d) exceeds ldc=%d",
                 name, p.coff, p.coff + p.Cout, p.ldc);
    AL3D_REQUIRE(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0,
                 "%s: in/wgt must be 16-byte aligned", name);
    return AL3D_OK;
}

extern "C" int al3d_conv2d_nhwc_f32(const float* in, const float* wgt, const float* scale,
                                    const float* shift, float* out, int B, int H, int W, int Cin,
                                    int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                    int relu, void* stream)
{
    ConvParams p;
    p.in = in; p.wgt = wgt; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ldc = ldc; p.coff = coff; p.relu = relu;
    AL3D_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0, "al3d_conv2d_nhwc_f32: bad geometry");
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    AL3D_REQUIRE(p.OH >= 1 && p.OW >= 1, "al3d_conv2d_nhwc_f32: empty output");
    int rc = conv_check(p, "al3d_conv2d_nhwc_f32");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(p.OW, CV_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, CV_TH);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)al3d_cdiv(Cout, CV_BN), 1);
    hipLaunchKernelGGL(conv2d_mfma_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_mfma_kernel<conv>");
    return AL3D_OK;
}

extern "C" int al3d_deconv2x2_nhwc_f32(const float* in, const float* wgt, const float* scale,
                                       const float* shift, float* out, int B, int H, int W, int Cin,
                                       int Cout, int ldc, int coff, int relu, void* stream)
{
    ConvParams p;
    p.in = in; p.wgt = wgt; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 2; p.stride = 2; p.pad = 0; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = 2 * H; p.OW = 2 * W;
    int rc = conv_check(p, "al3d_deconv2x2_nhwc_f32");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(W, CV_TW);
    p.tiles_y = (int)al3d_cdiv(H, CV_TH);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)al3d_cdiv(Cout, CV_BN), 4);
    hipLaunchKernelGGL(conv2d_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_mfma_kernel<deconv>");
    return AL3D_OK;
}

// ------------------------------------------------------------------ BEV embedding
// x.mean(-1).mean(-1) of the NCHW neck output == per channel: mean over W, then mean
// over H (feature_selector.py:68-71).  NHWC input [B,H,W,C] -> [B,C].  One workgroup
// per (image, 64-channel slab); rows are reduced left-to-right, then top-to-bottom.
// stage 1: one workgroup per image row (b, y): rowmean[b][y][c] = (sum_x x[b,y,x,c]) / W,
// lanes run over channels so every load is a full 1 KiB line; HBM-bound (reads the map once).
__global__ __launch_bounds__(256) void gap_rows_kernel(const float* __restrict__ x, int W, int C,
                                                       float* __restrict__ rowmean)
{
    const int64_t by = blockIdx.x;
    const float* row = x + by * W * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int xx = 0; xx < W; ++xx) s += row[(int64_t)xx * C + c];
        rowmean[by * C + c] = s / (float)W;
    }
}

// same reduction with 16-byte loads: a thread owns four consecutive channels (C % 4 == 0), so a row of
// C = 512 channels is one 128-lane load per pixel instead of two 256-lane ones; every channel is
// still summed left to right in its own accumulator (bit-identical to the scalar kernel).
__global__ __launch_bounds__(256) void gap_rows_vec4_kernel(const float* __restrict__ x, int W, int C, int rows,
                                                            float* __restrict__ rowmean)
{
    const int c4 = C >> 2;                                   // float4 lanes per pixel
    const int per_block = 256 / c4 > 0 ? 256 / c4 : 1;      // image rows handled by one workgroup
    const int sub = threadIdx.x / c4, lane = threadIdx.x % c4;
    const int64_t by = (int64_t)blockIdx.x * per_block + sub;
    if (sub >= per_block || by >= rows) return;
    const float4* row = reinterpret_cast<const float4*>(x + by * W * C) + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int xx = 0; xx < W; ++xx) {
        const float4 v = row[(int64_t)xx * c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float w = (float)W;
    reinterpret_cast<float4*>(rowmean + by * C)[lane] = make_float4(s.x / w, s.y / w, s.z / w, s.w / w);
}

// stage 2: out[b][c] = (sum_y rowmean[b][y][c]) / H, top to bottom
__global__ __launch_bounds__(64) void gap_cols_kernel(const float* __restrict__ rowmean, int H, int C,
                                                       float* __restrict__ out)
{
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;   // 64-thread blocks: more of them in flight
    if (c >= C) return;
    float s = 0.f;
#pragma unroll 16
    for (int y = 0; y < H; ++y) s += rowmean[((int64_t)b * H + y) * C + c];   // loads run ahead, adds stay in order
    out[(int64_t)b * C + c] = s / (float)H;
}

extern "C" int64_t al3d_gap_workspace_bytes(int B, int H, int C) { return (int64_t)B * H * C * 4; }

extern "C" int al3d_gap_nhwc_f32(const float* x, int B, int H, int W, int C, float* out,
                                 void* workspace, void* stream)
{
    AL3D_REQUIRE(x && out && workspace, "al3d_gap_nhwc_f32: null pointer");
    AL3D_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, "al3d_gap_nhwc_f32: bad shape");
    float* rowmean = (float*)workspace;
    if (C % 4 == 0 && C / 4 <= 256 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)workspace & 15) == 0) {
        const int per_block = 256 / (C / 4);
        hipLaunchKernelGGL(gap_rows_vec4_kernel, dim3((unsigned)al3d_cdiv((int64_t)B * H, per_block)), dim3(256), 0,
                           (hipStream_t)stream, x, W, C, B * H, rowmean);
    } else {
        hipLaunchKernelGGL(gap_rows_kernel, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, x, W, C,
                           rowmean);
    }
    hipLaunchKernelGGL(gap_cols_kernel, dim3((unsigned)al3d_cdiv(C, 64), (unsigned)B), dim3(64), 0,
                       (hipStream_t)stream, rowmean, H, C, out);
    AL3D_CHECK_LAUNCH("gap_kernel");
    return AL3D_OK;
}
