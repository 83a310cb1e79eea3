// Token-matrix kernels of the Swin-T image backbone (BASELINE configs[4]: BEVFusion camera+lidar,
// bevfusion/configs/nuscenes/det/transfusion/secfpn/camera+lidar/swint_v0p075/default.yaml -- embed 96,
// depths [2,2,6,2], heads [3,6,12,24], window 7; the module it configures is mmdet 2.20.0's SwinTransformer,
// absent from the reference tree: the published algorithm is restated, parity unpinned).
//
// A Swin block is  x += proj(window_attention(LN1(x)));  x += fc2(gelu(fc1(LN2(x)))).  On the matrix cores that is
// four skinny GEMMs over [tokens, C] matrices (K = 96 .. 3072) plus 49 x 49 attention per (window, head) -- here:
//
//   tok_layernorm_kernel         LayerNorm of token rows, optionally GATHERED through a row map (cyclic shift +
//                                window partition + padding for LN1; the 2x2 neighbourhood for patch merging) and
//                                written as "pair rows" (sp_rows.h): the f16 split (xh, xl') the f16x3 products
//                                multiply with, done once per activation instead of once per consuming tile
//   tok_linear_f16x3_kernel      [M, K] x [N, K]^T in fp32-class f16x3 arithmetic (conv2d_f16x3.hip: three f16
//                                products per MAC into one fp32 accumulator), both operands by LDS-DMA through a
//                                3-stage ring exactly as conv2d_f16x3_dma2_kernel, rows addressed directly;
//                                epilogue: bias, exact (erf) GELU, residual, row SCATTER (window order -> token
//                                order, padding rows dropped), f32 or pair rows out
//   tok_window_attention_kernel  two waves per (window, head), one per query tile: S^T = K (Q scale)^T on the matrix
//                                cores with both operands split (main + 2^-11 correction accumulators), + relative
//                                position bias + shifted-window region mask computed from the window's position (no
//                                mask tensor), softmax down the accumulator registers, O^T = V^T P^T with P taken
//                                straight from the accumulators as the B operand (no LDS round trip), pair rows out;
//                                reads qkv in window order or (token-order mode, embed 384 / 768) in the map's own order
//   tok_mlp_f16x3_kernel         embed 96: LN2 + fc1 + exact GELU + fc2 + residual in one kernel (hidden rows stay in
//                                registers, weights through an LDS-DMA ring)
//   tok_attn_block_f16x3_kernel  embed 96 / 192: LN1 + qkv + window attention + proj + residual in one kernel; 2 NH
//                                waves per window, K^T / V fragments through LDS, everything else in accumulators
//   tok_patch_embed_f16x3_kernel Conv2d(3, 96, 4, 4) + LayerNorm of the patch embedding in one kernel
//   tok_mha16_kernel (+ combine) the TransFusion decoder's multi-head attention (head dim 16)
//
// Nothing here calls a BLAS / MIOpen routine.  gfx950 only.
#include "al3d_common.h"
#include "sp_rows.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float tk_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void tk_lds_void;
typedef const __attribute__((address_space(1))) void tk_gbl_void;

#define TK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define TK_STAGE 16384      // bytes per ring stage: A 8 KB (128 rows x 16 channels f32 / pair) + B 8 KB
#define TK_BOFF 8192
#define TK_NS 3

__device__ __attribute__((aligned(256))) float g_tok_zero[64];     // stays zero: source of rows beyond M

template <int N> __device__ __forceinline__ void tk_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int tk_swz(int r) { return (r & 1) | (((r >> 3) & 1) << 1); }

__device__ __forceinline__ void tk_split(float x, _Float16& h, _Float16& l)
{
    // x is pinned to ONE rounded fp32 value first: with the producer's arithmetic visible (x = a * b), the compiler may
    // otherwise feed the residual below from the unrounded product while h rounds the rounded one -- at an f16 tie the
    // two then disagree by a whole f16 ulp (seen in the fused MLP: GELU output 0.27600098, low part with the wrong sign)
    asm volatile("" : "+v"(x));
    h = (_Float16)x;
    float hf = (float)h;
    asm volatile("" : "+v"(hf));                                     // the residual is taken against THIS h
    l = (_Float16)__builtin_fmaf(hf, -2048.0f, x * 2048.0f);         // (x - h) * 2^11 exactly
}
__device__ __forceinline__ void tk_split8(const float (&v)[8], f16x8& ph, f16x8& pl)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) { _Float16 a, b; tk_split(v[e], a, b); ph[e] = a; pl[e] = b; }
}
__device__ __forceinline__ void tk_split8v(const tk_f32x4& lo, const tk_f32x4& hi, f16x8& ph, f16x8& pl)
{
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    tk_split8(v, ph, pl);
}
__device__ __forceinline__ f16x8 tk_lift_down(const f16x8& wh) { return wh * (_Float16)0.00048828125f; }

// ------------------------------------------------------------------ LayerNorm over gathered rows
struct TokLnParams {
    const float* x;         // [rows_in][C]
    const int* rowmap;      // [rows_out * G] source row of each piece, -1 = absent; null = identity (row * G + piece)
    const float* gamma;     // [G * C]
    const float* beta;      // [G * C]
    float* out;             // [rows_out][G * C] f32 or pair rows
    int64_t rows_out;
    int C, G, zero_out, pair;
    float eps;
};

// LPR lanes per output row; a lane holds up to 3 groups of 8 consecutive channels in registers (G * C <= 24 * LPR).
// Two-pass statistics in fp32 (mean, then the centred second moment), biased variance, as torch.nn.LayerNorm.
// Absent pieces are zeros BEFORE the normalisation (patch merging pads the map, then normalises); with zero_out
// (G == 1) an absent row is a zero OUTPUT row instead (the window padding is applied to the normalised map).
template <int LPR>
__global__ __launch_bounds__(256) void tok_layernorm_kernel(TokLnParams p)
{
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < p.rows_out;
    const int Wd = p.G * p.C, ng = Wd >> 3;
    float v[3][8];
    bool absent_row = false;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int g = sub + t * LPR;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[t][e] = 0.f;
        if (!live || g >= ng) continue;
        const int ch = g << 3, piece = ch / p.C;
        const int64_t src = p.rowmap ? (int64_t)p.rowmap[row * p.G + piece] : row * p.G + piece;
        if (src < 0) { absent_row = true; continue; }
        const float* s = p.x + src * p.C + (ch - piece * p.C);
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        v[t][0] = a.x; v[t][1] = a.y; v[t][2] = a.z; v[t][3] = a.w;
        v[t][4] = b.x; v[t][5] = b.y; v[t][6] = b.z; v[t][7] = b.w;
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[t][e];
#pragma unroll
    for (int o = LPR >> 1; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)Wd;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (sub + t * LPR >= ng) continue;                 // registers of groups beyond the row are not part of it
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[t][e] - mean; sq += d * d; }
    }
#pragma unroll
    for (int o = LPR >> 1; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)Wd + p.eps);
    if (!live) return;
    const bool zero = p.zero_out && absent_row;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int g = sub + t * LPR;
        if (g >= ng) continue;
        const int ch = g << 3;
        float y[8];
        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + ch), g1 = *reinterpret_cast<const float4*>(p.gamma + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.beta + ch), b1 = *reinterpret_cast<const float4*>(p.beta + ch + 4);
        const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = zero ? 0.f : ((v[t][e] - mean) * rstd) * ga[e] + be[e];
        float* o = p.out + row * Wd + ch;
        if (p.pair) {
            uint4 hi, lo;
            sp_split8(y, hi, lo);
            *reinterpret_cast<uint4*>(o) = hi;
            *reinterpret_cast<uint4*>(o + 4) = lo;
        } else {
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
    }
}

extern "C" int al3d_tok_layernorm_f32(const float* x, const int* rowmap, int64_t rows_out, int C, int G, int zero_out,
                                      const float* gamma, const float* beta, float eps, int out_pair, float* out,
                                      void* stream)
{
    AL3D_REQUIRE(x && gamma && beta && out && rows_out >= 0, "al3d_tok_layernorm_f32: null pointer / negative rows");
    AL3D_REQUIRE(C >= 8 && C % 8 == 0 && (G == 1 || G == 4) && G * C <= 1536,
                 "al3d_tok_layernorm_f32: C=%d (multiple of 8), G=%d (1 or 4), G*C <= 1536", C, G);
    AL3D_REQUIRE(!zero_out || G == 1, "al3d_tok_layernorm_f32: zero_out needs G == 1");
    AL3D_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                 "al3d_tok_layernorm_f32: pointers must be 16-byte aligned");
    if (rows_out == 0) return AL3D_OK;
    TokLnParams p{x, rowmap, gamma, beta, out, rows_out, C, G, zero_out, out_pair, eps};
    if (G * C <= 384)
        hipLaunchKernelGGL(tok_layernorm_kernel<16>, dim3((unsigned)al3d_cdiv(rows_out, 16)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(tok_layernorm_kernel<64>, dim3((unsigned)al3d_cdiv(rows_out, 4)), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("tok_layernorm_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ patch rows of the 4 x 4 / stride 4 embedding
// img [B][H][W][3] f32 channels-last -> rows [B * TH * TW][48] (pair rows or f32), row = one 4 x 4 patch in the order
// k = (ky * 4 + kx) * 3 + c; pixels beyond H / W are zero (the reference pads the image to a multiple of the patch).
// One lane per (row, group of 8 k).
__global__ __launch_bounds__(256) void tok_patch_rows_kernel(const float* __restrict__ img, int B, int H, int W, int TH, int TW,
                                                             int pair, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)B * TH * TW;
    if (t >= rows * 6) return;
    const int g = (int)(t % 6);
    int64_t row = t / 6;
    const int tx = (int)(row % TW), ty = (int)((row / TW) % TH), b = (int)(row / ((int64_t)TW * TH));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e, ky = k / 12, r = k - 12 * ky, kx = r / 3, ch = r - 3 * kx;
        const int y = 4 * ty + ky, x = 4 * tx + kx;
        v[e] = (y < H && x < W) ? img[(((int64_t)b * H + y) * W + x) * 3 + ch] : 0.f;
    }
    float* o = out + row * 48 + 8 * g;
    if (pair) {
        uint4 hi, lo;
        sp_split8(v, hi, lo);
        *reinterpret_cast<uint4*>(o) = hi;
        *reinterpret_cast<uint4*>(o + 4) = lo;
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

extern "C" int al3d_tok_patch_rows_f32(const float* img, int B, int H, int W, int out_pair, float* out, void* stream)
{
    AL3D_REQUIRE(img && out && B >= 1 && H >= 1 && W >= 1, "al3d_tok_patch_rows_f32: bad arguments");
    AL3D_REQUIRE(((uintptr_t)out & 15) == 0, "al3d_tok_patch_rows_f32: out must be 16-byte aligned");
    const int TH = (H + 3) / 4, TW = (W + 3) / 4;
    const int64_t n = (int64_t)B * TH * TW * 6;
    hipLaunchKernelGGL(tok_patch_rows_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img, B, H, W, TH,
                       TW, out_pair, out);
    AL3D_CHECK_LAUNCH("tok_patch_rows_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ token GEMM, f16x3, both operands by LDS-DMA
struct TokGemmParams {
    const float* a;         // [M][K] f32 rows or pair rows
    const _Float16* wgt;    // al3d_pack_f16x3_dma image (taps = 1): [ceil(N/128)][K/16][2 planes][128][16]
    const float* scale;     // [N]: 2^-s of the weight split (required)
    const float* bias;      // [N] or null
    const float* residual;  // [*][ldr] f32, indexed by OUTPUT row, or null
    const int* rowmap;      // [M] output row of GEMM row m (-1: dropped) or null
    float* out;             // [*][ldc]
    int M, K, N, ldc, ldr, act;
    int ntiles, nblocks;
};

// same XCD-aware (row tile, column block) order as conv2d_f16x3.hip's f3_tile_of_block
__device__ __forceinline__ bool tk_tile_of_block(const TokGemmParams& p, int& tile, int& nblk)
{
    const int id = blockIdx.x, span = 8 * p.nblocks;
    const int grp = id / span, rem = id - grp * span;
    nblk = rem >> 3;
    tile = grp * 8 + (rem & 7);
    return tile < p.ntiles;
}

// erf to fp32 rounding level, branch-free: the two minimax pieces of N. Juffa's single-precision erf (x + x P(x^2) below
// 475/512, 1 - exp(Q(|x|)) above; each < 1 ulp with an exact exp) are both evaluated and one is selected -- the library erff
// costs ~45 vector instructions and a divergent branch per element, and the GELU epilogues are bound by exactly that.
// exp through v_exp_f32 (2^x): its argument is <= -0.9, so the result is <= 0.41 and the error it adds to 1 - exp stays
// below 1e-7.  Measured against float64 over [-8, 8]: see tests/test_swin_gpu.py::test_gelu_epilogue_accuracy.
__device__ __forceinline__ float tk_erf(float a)
{
    const float t = fabsf(a), s = a * a;
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    float big = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f);
    big = __builtin_copysignf(big, a);
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    q = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? big : q;
}

__device__ __forceinline__ float tk_gelu(float v)
{
    return (v * 0.5f) * (1.0f + tk_erf(v * 0.70710678118654752440f));
}

struct TkOps {
    f16x8 ah[2], al[2], wh[2], wl[2], wd[2];
};

// IO bit 0: A rows are pair rows; bit 1: write pair rows.  Tile 128 rows x 128 columns x 16 channels per step,
// four waves of 64 x 64; ring, fragment offsets, asm block and product order of conv2d_f16x3_dma2_kernel.
template <int IO>
__global__ __launch_bounds__(256, 2) void tok_linear_f16x3_kernel(TokGemmParams p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[TK_NS * TK_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int tile, nblk;
    if (!tk_tile_of_block(p, tile, nblk)) return;      // padding block of the last group (uniform)
    const int m0 = tile * 128, n0 = nblk * 128;
    const int total = p.K >> 4;
    const unsigned smem_base = (unsigned)(size_t)(tk_lds_void*)smem;

    // DMA side: wave w fetches rows 32 w .. 32 w + 31 and quarter w of the weight tile; lane (jg, sg) of piece i
    // fetches 16-byte chunk sg ^ f(r) of row r = 2 jg + i (source-side swizzle: the LDS side of a DMA is lane-linear)
    const int jg = lane >> 2, sg = lane & 3;
    const char* abase[2];
    unsigned ainc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 2 * jg + i, m = m0 + 32 * wave + r;
        const int cg = (sg ^ tk_swz(r)) * 4;
        const bool ok = m < p.M;
        abase[i] = reinterpret_cast<const char*>(ok ? p.a + (int64_t)m * p.K + cg : g_tok_zero + cg);
        ainc[i] = ok ? 64u : 0u;
    }
    const char* wsrc = reinterpret_cast<const char*>(p.wgt) + (int64_t)nblk * total * 8192 + wave * 2048 + lane * 16;
    int lchunk = 0;
    auto issue = [&](int stage) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + stage * TK_STAGE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((tk_gbl_void*)(abase[i] + (size_t)lchunk * ainc[i]),
                                             (tk_lds_void*)(size_t)(dst + i * 1024), 16, 0, 0);
        const char* ws = wsrc + (int64_t)lchunk * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((tk_gbl_void*)(ws + i * 1024), (tk_lds_void*)(size_t)(dst + TK_BOFF + i * 1024),
                                             16, 0, 0);
        if (lchunk + 1 < total) ++lchunk;              // steps past the end re-fetch the last one: 4 DMAs per step, always
    };

    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }

    const int fr = lane & 31, fh = lane >> 5;
    const unsigned offA0 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh) ^ tk_swz(fr)) * 16));
    const unsigned offA1 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh + 1) ^ tk_swz(fr)) * 16));
    const unsigned offB = (unsigned)(TK_BOFF + (wn * 64 + fr) * 32 + ((fh ^ ((fr >> 3) & 1)) * 16));

    TkOps A, B;
    constexpr bool in_pair = (IO & 1) != 0;
    auto finish = [&](TkOps& o, const tk_f32x4& r0l, const tk_f32x4& r0h, const tk_f32x4& r1l, const tk_f32x4& r1h) {
        if constexpr (in_pair) {                         // the 32 bytes of a fragment ARE (xh[8], xl'[8])
            o.ah[0] = __builtin_bit_cast(f16x8, r0l); o.al[0] = __builtin_bit_cast(f16x8, r0h);
            o.ah[1] = __builtin_bit_cast(f16x8, r1l); o.al[1] = __builtin_bit_cast(f16x8, r1h);
        } else {
            tk_split8v(r0l, r0h, o.ah[0], o.al[0]);
            tk_split8v(r1l, r1h, o.ah[1], o.al[1]);
        }
        o.wd[0] = tk_lift_down(o.wh[0]);
        o.wd[1] = tk_lift_down(o.wh[1]);
    };

#pragma unroll
    for (int s = 0; s < TK_NS; ++s) issue(s);
    {
        tk_wait_vm<4 * (TK_NS - 1)>();                 // stage 0
        __builtin_amdgcn_s_barrier();
        tk_f32x4 r0l, r0h, r1l, r1h;
        asm volatile("ds_read_b128 %0, %8\n\t"
                     "ds_read_b128 %1, %9\n\t"
                     "ds_read_b128 %2, %8 offset:2048\n\t"
                     "ds_read_b128 %3, %9 offset:2048\n\t"
                     "ds_read_b128 %4, %10\n\t"
                     "ds_read_b128 %5, %10 offset:4096\n\t"
                     "ds_read_b128 %6, %10 offset:1024\n\t"
                     "ds_read_b128 %7, %10 offset:5120\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0l), "=&v"(r0h), "=&v"(r1l), "=&v"(r1h), "=&v"(A.wh[0]), "=&v"(A.wl[0]), "=&v"(A.wh[1]),
                       "=&v"(A.wl[1])
                     : "v"(smem_base + offA0), "v"(smem_base + offA1), "v"(smem_base + offB) : "memory");
        finish(A, r0l, r0h, r1l, r1h);
    }
    int nstage = 1 % TK_NS, fill = 0;
    auto step = [&](TkOps& cur, TkOps& nxt) {
        tk_wait_vm<4 * (TK_NS - 2)>();                 // this wave's share of the next stage has landed ...
        __builtin_amdgcn_s_barrier();                  // ... everyone's has; every wave holds the current stage in registers
        issue(fill);                                   // -> the current stage's buffer
        const unsigned sb = smem_base + nstage * TK_STAGE;
        tk_f32x4 r0l, r0h, r1l, r1h;
        asm volatile("s_nop 1\n\t"
                     "ds_read_b128 %0, %22\n\t"
                     "ds_read_b128 %1, %23\n\t"
                     "v_mfma_f32_32x32x16_f16 %8, %12, %16, %8\n\t"
                     "v_mfma_f32_32x32x16_f16 %9, %12, %19, %9\n\t"
                     "ds_read_b128 %2, %22 offset:2048\n\t"
                     "ds_read_b128 %3, %23 offset:2048\n\t"
                     "v_mfma_f32_32x32x16_f16 %10, %14, %16, %10\n\t"
                     "v_mfma_f32_32x32x16_f16 %11, %14, %19, %11\n\t"
                     "ds_read_b128 %4, %24\n\t"
                     "ds_read_b128 %5, %24 offset:4096\n\t"
                     "v_mfma_f32_32x32x16_f16 %8, %13, %17, %8\n\t"
                     "v_mfma_f32_32x32x16_f16 %9, %13, %20, %9\n\t"
                     "ds_read_b128 %6, %24 offset:1024\n\t"
                     "ds_read_b128 %7, %24 offset:5120\n\t"
                     "v_mfma_f32_32x32x16_f16 %10, %15, %17, %10\n\t"
                     "v_mfma_f32_32x32x16_f16 %11, %15, %20, %11\n\t"
                     "v_mfma_f32_32x32x16_f16 %8, %13, %18, %8\n\t"
                     "v_mfma_f32_32x32x16_f16 %9, %13, %21, %9\n\t"
                     "v_mfma_f32_32x32x16_f16 %10, %15, %18, %10\n\t"
                     "v_mfma_f32_32x32x16_f16 %11, %15, %21, %11\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0l), "=&v"(r0h), "=&v"(r1l), "=&v"(r1h), "=&v"(nxt.wh[0]), "=&v"(nxt.wl[0]), "=&v"(nxt.wh[1]),
                       "=&v"(nxt.wl[1]), "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11)
                     : "v"(cur.al[0]), "v"(cur.ah[0]), "v"(cur.al[1]), "v"(cur.ah[1]),                    // 12..15
                       "v"(cur.wd[0]), "v"(cur.wl[0]), "v"(cur.wh[0]), "v"(cur.wd[1]), "v"(cur.wl[1]), "v"(cur.wh[1]),   // 16..21
                       "v"(sb + offA0), "v"(sb + offA1), "v"(sb + offB)                                   // 22..24
                     : "memory");
        finish(nxt, r0l, r0h, r1l, r1h);
        nstage = nstage + 1 == TK_NS ? 0 : nstage + 1;
        fill = fill + 1 == TK_NS ? 0 : fill + 1;
    };
    for (int s = 0; s < total; s += 2) {
        step(A, B);
        if (s + 1 < total) step(B, A);
    }
    tk_wait_vm<0>();                                   // the tail's dummy requests must not outlive the workgroup's LDS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the VALU reads them

    // epilogue: scale / bias / GELU in the C layout, then each 32-row x 64-column half of the wave's tile goes
    // through a wave-private LDS scratch (the ring, free after the barrier) so that a lane holds consecutive channels
    // of a row: residual loads and stores are 16-byte pieces
    const f32x16* accp[2][2] = {{&c00, &c01}, {&c10, &c11}};
    __syncthreads();
    constexpr int SP = 68;
    float* scr = reinterpret_cast<float*>(smem) + wave * (32 * SP);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fr;
            const bool nok = n < p.N;
            const float sc = nok ? p.scale[n] : 0.f;
            const float sh = (nok && p.bias) ? p.bias[n] : 0.0f;
            const f32x16& acc = *accp[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[r] * sc + sh;
                if (p.act == 1) v = tk_gelu(v);
                else if (p.act == 2) v = v <= 0.f ? 0.f : v;         // NaN propagates, like torch.relu
                scr[ml * SP + j * 32 + fr] = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if constexpr ((IO & 2) != 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = lane + 64 * q, pl = t >> 3, g = t & 7;
                const int m = m0 + wm * 64 + i * 32 + pl;
                const int n = n0 + wn * 64 + g * 8;
                if (m >= p.M || n >= p.N) continue;
                const int orow = p.rowmap ? p.rowmap[m] : m;
                if (orow < 0) continue;
                const float4 a = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8);
                const float4 b4 = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8 + 4);
                float v[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
                if (p.residual) {
                    const float* rs = p.residual + (int64_t)orow * p.ldr + n;
                    const float4 ra = *reinterpret_cast<const float4*>(rs), rb = *reinterpret_cast<const float4*>(rs + 4);
                    v[0] += ra.x; v[1] += ra.y; v[2] += ra.z; v[3] += ra.w; v[4] += rb.x; v[5] += rb.y; v[6] += rb.z; v[7] += rb.w;
                }
                uint4 hi, lo;
                sp_split8(v, hi, lo);
                float* o = p.out + (int64_t)orow * p.ldc + n;
                *reinterpret_cast<uint4*>(o) = hi;
                *reinterpret_cast<uint4*>(o + 4) = lo;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = lane + 64 * q, pl = t >> 4, g = t & 15;
                const int m = m0 + wm * 64 + i * 32 + pl;
                const int n = n0 + wn * 64 + g * 4;
                if (m >= p.M || n >= p.N) continue;
                const int orow = p.rowmap ? p.rowmap[m] : m;
                if (orow < 0) continue;
                float4 v = *reinterpret_cast<const float4*>(scr + pl * SP + g * 4);
                if (p.residual) {
                    const float4 rs = *reinterpret_cast<const float4*>(p.residual + (int64_t)orow * p.ldr + n);
                    v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                }
                *reinterpret_cast<float4*>(p.out + (int64_t)orow * p.ldc + n) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int al3d_tok_linear_f16x3(const float* a, int a_pair, const void* wgt_image, const float* scale,
                                     const float* bias, int64_t M, int K, int N, int act, const float* residual,
                                     int ldr, const int* rowmap, float* out, int ldc, int out_pair, void* stream)
{
    AL3D_REQUIRE(a && wgt_image && scale && out, "al3d_tok_linear_f16x3: null pointer (scale carries the weight exponent and is required)");
    AL3D_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) - 128 && K >= 16 && K % 16 == 0 && N >= 1, "al3d_tok_linear_f16x3: bad shape M=%lld K=%d N=%d",
                 (long long)M, K, N);
    const int q = out_pair ? 8 : 4;
    AL3D_REQUIRE(N % q == 0 && ldc % q == 0 && ldc >= N, "al3d_tok_linear_f16x3: N=%d, ldc=%d must be multiples of %d, ldc >= N", N, ldc, q);
    AL3D_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= N), "al3d_tok_linear_f16x3: ldr=%d must be a multiple of 4 and >= N", ldr);
    AL3D_REQUIRE(act >= 0 && act <= 2, "al3d_tok_linear_f16x3: act = 0 (none), 1 (GELU) or 2 (ReLU)");
    AL3D_REQUIRE((((uintptr_t)a | (uintptr_t)wgt_image | (uintptr_t)out | (uintptr_t)residual) & 15) == 0,
                 "al3d_tok_linear_f16x3: a / wgt / out / residual must be 16-byte aligned");
    if (M == 0) return AL3D_OK;
    TokGemmParams p;
    p.a = a; p.wgt = (const _Float16*)wgt_image; p.scale = scale; p.bias = bias; p.residual = residual; p.rowmap = rowmap;
    p.out = out; p.M = (int)M; p.K = K; p.N = N; p.ldc = ldc; p.ldr = ldr; p.act = act;
    p.ntiles = (int)al3d_cdiv(M, 128); p.nblocks = (int)al3d_cdiv(N, 128);
    const dim3 grid((unsigned)(al3d_cdiv(p.ntiles, 8) * 8 * p.nblocks));
    hipStream_t s = (hipStream_t)stream;
    switch ((a_pair ? 1 : 0) | (out_pair ? 2 : 0)) {
    case 0: hipLaunchKernelGGL(tok_linear_f16x3_kernel<0>, grid, dim3(256), 0, s, p); break;
    case 1: hipLaunchKernelGGL(tok_linear_f16x3_kernel<1>, grid, dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(tok_linear_f16x3_kernel<2>, grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(tok_linear_f16x3_kernel<3>, grid, dim3(256), 0, s, p); break;
    }
    AL3D_CHECK_LAUNCH("tok_linear_f16x3_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ fused MLP half of a Swin block (C = 96)
// x += fc2(gelu(fc1(LN2(x)))) as ONE kernel: the [tokens, 4 C] hidden activation -- 8 C of the 26 C floats a block moves
// per token when its six GEMM / LN launches run separately, and these layers are bandwidth-bound -- never leaves the
// registers.  Everything is computed TRANSPOSED so that one GEMM's accumulator is the next one's operand as it stands:
// a wave owns 32 tokens (the n dimension of every product); LN2 runs on the lane's half row, its split (xh, xl') IS the B
// operand of H^T = W1 xn^T; a 32 x 32 tile of H^T (hidden units down the accumulator registers, tokens across the lanes)
// gets bias + exact GELU + the f16 split in place and IS the B operand of out^T += W2 H^T -- the hidden index inside a
// 16-chunk is then the accumulator's row order (e&3) + 8 (e>>2) + 4 fh + 16 q, which the host bakes into the W2 image.
// The weights are the A operands: per hidden tile t the image holds W1's K/16 x (wh, wl) fragments and W2's
// (C/32) x 2 x (wh, wl) fragments in MFMA lane order (C/4 KB), brought in by LDS-DMA into a two-stage ring shared by the
// workgroup's four waves, one barrier per hidden tile.  f16x3 arithmetic as everywhere (three products per MAC:
// wd xl' + wl xh + wh xh, wd = wh 2^-11).  The result leaves through a per-wave LDS transposition (32 tokens x 32 channels
// at a time) so that residual loads and stores are 16-byte pieces of 128-byte row segments.
struct TokMlpParams {
    float* x;                // [T][C] residual stream, updated in place
    const _Float16* image;   // [NT][C/4 KB]: per hidden tile W1 fragments [C/16][2][64][8], then W2 fragments [C/32][2][2][64][8]
    const float* gamma;      // [C] LN2
    const float* beta;       // [C]
    const float* bias1;      // [32 NT]
    const float* bias2;      // [C]
    float scale1, scale2, eps;     // 2^-s of the two weight splits
    int64_t T;
    int NT;                  // hidden / 32
};

// NW waves per workgroup (32 tokens each) share a ring of NS stages; a tile is requested NS - 1 tiles ahead
template <int C, int NW, int NS>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void tok_mlp_f16x3_kernel(TokMlpParams p)
{
    constexpr int KC = C / 16, U = C / 32, TILE = C * 256, W2OFF = KC * 2048;
    extern __shared__ __attribute__((aligned(1024))) unsigned char mlp_smem[];
    constexpr int NTHR = NW * 64, DMA_ROUND = NW * 1024, RPT = TILE / DMA_ROUND;      // DMA instructions per tile and wave
    unsigned char* ring = mlp_smem;                                     // NS x TILE
    float* b1s = reinterpret_cast<float*>(mlp_smem + NS * TILE);        // [32 NT]
    float* gs = b1s + 32 * p.NT;                                        // gamma [C] | beta [C] | bias2 [C]
    float* scr = gs + 3 * C + (threadIdx.x >> 6) * (32 * 33);           // per wave: 32 tokens x 33
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring_base = (unsigned)(size_t)(tk_lds_void*)ring;

    auto issue = [&](int t) {                                           // hidden tile t -> ring stage t % NS
        const char* src = reinterpret_cast<const char*>(p.image) + (int64_t)t * TILE + wave * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (t % NS) * TILE + wave * 1024);
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            __builtin_amdgcn_global_load_lds((tk_gbl_void*)(src + r * DMA_ROUND), (tk_lds_void*)(size_t)(dst + r * DMA_ROUND), 16, 0, 0);
    };
    for (int i = tid; i < 32 * p.NT; i += NTHR) b1s[i] = p.bias1[i];
    for (int i = tid; i < C; i += NTHR) { gs[i] = p.gamma[i]; gs[C + i] = p.beta[i]; gs[2 * C + i] = p.bias2[i]; }
    __syncthreads();

    // ---- LN2 of the wave's 32 tokens: lane (fr, fh) holds channels kc * 16 + fh * 8 + e of token fr
    const int64_t tok = (int64_t)blockIdx.x * (NW * 32) + wave * 32 + fr;
    const bool live = tok < p.T;
    const float* xrow = p.x + (live ? tok : 0) * C + fh * 8;
    float xv[KC][8];
    float sum = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (live) { a = *reinterpret_cast<const float4*>(xrow + kc * 16); b = *reinterpret_cast<const float4*>(xrow + kc * 16 + 4); }
        xv[kc][0] = a.x; xv[kc][1] = a.y; xv[kc][2] = a.z; xv[kc][3] = a.w;
        xv[kc][4] = b.x; xv[kc][5] = b.y; xv[kc][6] = b.z; xv[kc][7] = b.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[kc][e];
    }
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[kc][e] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.0f / sqrtf(sq / (float)C + p.eps);
    f16x8 xh[KC], xl[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = kc * 16 + fh * 8 + e;
            v[e] = (xv[kc][e] - mean) * rstd * gs[ch] + gs[C + ch];
        }
        tk_split8(v, xh[kc], xl[kc]);
    }

    f32x16 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    // the x loads above have been consumed (their data was used): only DMA requests are in flight from here on
#pragma unroll
    for (int t0 = 0; t0 < NS - 1; ++t0) if (t0 < p.NT) issue(t0);
    for (int t = 0; t < p.NT; ++t) {
        // tiles t .. t + NS - 2 are in flight (fewer at the end): this wave's share of tile t has landed when at most the
        // younger tiles' requests remain
        if (t + NS - 2 < p.NT) tk_wait_vm<(NS - 2) * RPT>(); else tk_wait_vm<0>();
        __syncthreads();                                   // ... everyone's has, and nobody reads tile t - 1 any more
        if (t + NS - 1 < p.NT) issue(t + NS - 1);
        const unsigned char* st = ring + (t % NS) * TILE + lane * 16;
        f32x16 h;
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (kc * 2) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (kc * 2 + 1) * 1024);
            const f16x8 wd = tk_lift_down(wh);
            h = TK_MFMA(wd, xl[kc], h);
            h = TK_MFMA(wl, xh[kc], h);
            h = TK_MFMA(wh, xh[kc], h);
        }
        // bias + GELU + split: registers 8 q .. 8 q + 7 are B-operand chunk q (hidden unit (e&3) + 8 (e>>2) + 4 fh + 16 q)
        f16x8 hh[2], hl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(b1s + 32 * t + 16 * q + 8 * g + 4 * fh);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = tk_gelu(h[8 * q + 4 * g + e] * p.scale1 + bb[e]);
            }
            tk_split8(v, hh[q], hl[q]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(st + W2OFF + ((u * 2 + q) * 2) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(st + W2OFF + ((u * 2 + q) * 2 + 1) * 1024);
                const f16x8 wd = tk_lift_down(wh);
                acc[u] = TK_MFMA(wd, hl[q], acc[u]);
                acc[u] = TK_MFMA(wl, hh[q], acc[u]);
                acc[u] = TK_MFMA(wh, hh[q], acc[u]);
            }
    }

    // ---- out^T tiles -> rows: 32 tokens x 32 channels at a time through the wave's scratch
    const int64_t tok0 = (int64_t)blockIdx.x * (NW * 32) + wave * 32;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * fh;             // channel inside the tile; token = fr
            scr[fr * 33 + cl] = acc[u][r] * p.scale2 + gs[2 * C + 32 * u + cl];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q, tk = i >> 3, g = i & 7;        // token tk, channels 4 g .. 4 g + 3
            if (tok0 + tk < p.T) {
                float* o = p.x + (tok0 + tk) * C + 32 * u + 4 * g;
                float4 v = *reinterpret_cast<const float4*>(o);
                v.x += scr[tk * 33 + 4 * g]; v.y += scr[tk * 33 + 4 * g + 1];
                v.z += scr[tk * 33 + 4 * g + 2]; v.w += scr[tk * 33 + 4 * g + 3];
                *reinterpret_cast<float4*>(o) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the same MLP half at C = 192 on the 16 x 16 x 32 product: a wave owns SIXTEEN tokens.  At 32 tokens per wave the
// operands of the kernel above (xn^T 96 + out^T 96 registers + a hidden tile) need ~270 registers; with 16-token tiles they
// are 48 + 48 and three waves per SIMD fit (<= 168).  Lane (lr = lane & 15, lq = lane >> 4): a B fragment holds k-block lq
// (8 consecutive k) of column lr = token lr, an A fragment k-block lq of row lr, an accumulator rows 4 lq .. 4 lq + 3 of
// column lr.  So the x row pieces a lane loads (channels 32 ks + 8 lq .. + 7 of token lr) ARE fc1's B fragments after LN + split;
// the two 16-row tiles of a 32-unit hidden group, after bias + GELU + split, ARE fc2's B fragment in the k order
// (e < 4: unit 4 lq + e, e >= 4: unit 16 + 4 lq + e - 4) the host bakes into the W2 image; and out^T leaves as one float4 per
// lane and 16-channel tile (channels 16 u + 4 lq .. + 3 of token lr): no LDS transposition.  Weights: per hidden group 24 KB
// of W1 fragments [j][ks][plane][64][8] and 24 KB of W2 fragments [u][plane][64][8], one after the other through a four-stage
// LDS-DMA ring shared by the workgroup's twelve waves (192 tokens; one barrier per half group).
template <int C>
__global__ __launch_bounds__(768, 3) void tok_mlp16_f16x3_kernel(TokMlpParams p)
{
    constexpr int KS = C / 32, U = C / 16, HALF = C * 128, NW = 12, NS = 4;        // HALF: bytes of one ring stage
    constexpr int DPW = HALF / (NW * 1024);                                        // DMA instructions per stage and wave
    static_assert(HALF % (NW * 1024) == 0, "a stage must be whole DMA rounds of the workgroup");
    extern __shared__ __attribute__((aligned(1024))) unsigned char mlp_smem[];
    unsigned char* ring = mlp_smem;                                     // NS x HALF
    float* b1s = reinterpret_cast<float*>(mlp_smem + NS * HALF);        // [32 NT]
    float* gs = b1s + 32 * p.NT;                                        // gamma [C] | beta [C] | bias2 [C]
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring_base = (unsigned)(size_t)(tk_lds_void*)ring;
    const int nstage = 2 * p.NT;

    auto issue = [&](int st) {                                          // stage st (clamped) -> ring slot st % NS
        const int sc = st < nstage ? st : nstage - 1;                   // past the end: a harmless re-fetch into a finished slot
        const char* src = reinterpret_cast<const char*>(p.image) + (int64_t)sc * HALF + wave * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (st % NS) * HALF + wave * 1024);
#pragma unroll
        for (int r = 0; r < DPW; ++r)
            __builtin_amdgcn_global_load_lds((tk_gbl_void*)(src + r * (NW * 1024)), (tk_lds_void*)(size_t)(dst + r * (NW * 1024)), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue(st);                     // the ring fills while the rows are loaded and normalised
    for (int i = tid; i < 32 * p.NT; i += 768) b1s[i] = p.bias1[i];
    for (int i = tid; i < C; i += 768) { gs[i] = p.gamma[i]; gs[C + i] = p.beta[i]; gs[2 * C + i] = p.bias2[i]; }
    __syncthreads();

    // ---- LN2 of the wave's 16 tokens: lane (lr, lq) holds channels 32 ks + 8 lq + e of token lr
    const int64_t tok = (int64_t)blockIdx.x * (NW * 16) + wave * 16 + lr;
    const bool live = tok < p.T;
    float* xrow = p.x + (live ? tok : 0) * C;
    // three passes over the lane's 6 x 32 bytes (sum; squared deviations; normalise + split), each re-reading them (L2 hits)
    // instead of holding 48 values beside the 48 registers of the split: held, the prologue spilled 34 registers per lane --
    // 150 MB of scratch writes and as many reads per launch (PMC WRITE_SIZE 346 MB against 208 MB of output)
    auto ld8 = [&](int ks, float (&v)[8]) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (live) { a = *reinterpret_cast<const float4*>(xrow + 32 * ks + 8 * lq); b = *reinterpret_cast<const float4*>(xrow + 32 * ks + 8 * lq + 4); }
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        ld8(ks, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[e];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        ld8(ks, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; sq += d * d; }
    }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.0f / sqrtf(sq / (float)C + p.eps);
    f16x8 xh[KS], xl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        ld8(ks, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = 32 * ks + 8 * lq + e;
            v[e] = (v[e] - mean) * rstd * gs[ch] + gs[C + ch];
        }
        tk_split8(v, xh[ks], xl[ks]);
    }

    __builtin_amdgcn_sched_barrier(0);                       // the accumulators start to live only here (the prologue spilled)
    tk_f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = tk_f32x4{0.f, 0.f, 0.f, 0.f};

    // the x loads above have been consumed (and with them, in order, the ring's first stages): from here on only the DMA
    // requests of the loop are in flight (DPW per stage and wave)
    f16x8 hh, hl;
    for (int g = 0; g < p.NT; ++g) {
        // ---- first half: H^T = W1[group g] xn^T, two 16-unit tiles, + bias + GELU + split.  Fragments are requested one
        // ahead of their products (the LDS round trip of fragment i + 1 runs under the products of fragment i); the
        // correction product wh xl' goes to its own accumulator and is scaled once (2^-11) instead of scaling every wh.
        // (Measured and not kept: a third of the waves one step behind the others, so that the GELU phases of a SIMD's
        // three waves do not coincide -- 0.78 against 0.65 ms.)
        tk_wait_vm<(NS - 2) * DPW>();                        // this wave's share of stage 2 g has landed ...
        __syncthreads();                                     // ... everyone's has, and nobody reads stage 2 g - 1 any more
        issue(2 * g + NS - 1);
        {
            const unsigned char* st = ring + ((2 * g) % NS) * HALF + lane * 16;
            tk_f32x4 h[2], hc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { h[j] = tk_f32x4{0.f, 0.f, 0.f, 0.f}; hc[j] = h[j]; }
            f16x8 wh[2], wl[2];
            wh[0] = *reinterpret_cast<const f16x8*>(st);
            wl[0] = *reinterpret_cast<const f16x8*>(st + 1024);
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                if (i + 1 < 2 * KS) {
                    wh[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(st + ((i + 1) * 2) * 1024);
                    wl[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(st + ((i + 1) * 2 + 1) * 1024);
                }
                const int j = i / KS, ks = i % KS;
                hc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i & 1], xl[ks], hc[j], 0, 0, 0);
                h[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i & 1], xh[ks], h[j], 0, 0, 0);
                h[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i & 1], xh[ks], h[j], 0, 0, 0);
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(b1s + 32 * g + 16 * j + 4 * lq);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = tk_gelu((h[j][e] + hc[j][e] * 0.00048828125f) * p.scale1 + bb[e]);
            }
            tk_split8(v, hh, hl);
        }
        // ---- second half: out^T += W2[:, group g] H^T
        tk_wait_vm<(NS - 2) * DPW>();
        __syncthreads();
        issue(2 * g + NS);
        {
            const unsigned char* st = ring + ((2 * g + 1) % NS) * HALF + lane * 16;
            f16x8 wh[2], wl[2];
            wh[0] = *reinterpret_cast<const f16x8*>(st);
            wl[0] = *reinterpret_cast<const f16x8*>(st + 1024);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u + 1 < U) {
                    wh[(u + 1) & 1] = *reinterpret_cast<const f16x8*>(st + ((u + 1) * 2) * 1024);
                    wl[(u + 1) & 1] = *reinterpret_cast<const f16x8*>(st + ((u + 1) * 2 + 1) * 1024);
                }
                const f16x8 wd = tk_lift_down(wh[u & 1]);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd, hl, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u & 1], hh, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u & 1], hh, acc[u], 0, 0, 0);
            }
        }
    }
    tk_wait_vm<0>();                                         // the tail's dummy requests must not outlive the workgroup's LDS

    // ---- out^T tiles are row pieces already: channels 16 u + 4 lq .. + 3 of token lr
    if (live) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float* o = xrow + 16 * u + 4 * lq;
            const float4 b2 = *reinterpret_cast<const float4*>(gs + 2 * C + 16 * u + 4 * lq);
            float4 v = *reinterpret_cast<const float4*>(o);
            v.x += acc[u][0] * p.scale2 + b2.x; v.y += acc[u][1] * p.scale2 + b2.y;
            v.z += acc[u][2] * p.scale2 + b2.z; v.w += acc[u][3] * p.scale2 + b2.w;
            *reinterpret_cast<float4*>(o) = v;
        }
    }
}

extern "C" int64_t al3d_tok_mlp_image_bytes(int C, int hidden) { return (int64_t)(hidden / 32) * C * 256; }

extern "C" int al3d_tok_mlp_f16x3(float* x, int64_t T, int C, int hidden, const float* gamma, const float* beta, float eps,
                                  const void* image, float scale1, const float* bias1, float scale2, const float* bias2,
                                  void* stream)
{
    if (T == 0) return AL3D_OK;                      // an empty row matrix has no storage
    AL3D_REQUIRE(x && gamma && beta && image && bias1 && bias2, "al3d_tok_mlp_f16x3: null pointer");
    AL3D_REQUIRE(C == 96 || C == 192, "al3d_tok_mlp_f16x3: built for C = 96 (32-token waves) and C = 192 (16-token waves), got %d", C);
    AL3D_REQUIRE(hidden >= 32 && hidden % 32 == 0 && T >= 0, "al3d_tok_mlp_f16x3: hidden must be a multiple of 32");
    AL3D_REQUIRE((((uintptr_t)x | (uintptr_t)image) & 15) == 0, "al3d_tok_mlp_f16x3: x / image must be 16-byte aligned");
    TokMlpParams p;
    p.x = x; p.image = (const _Float16*)image; p.gamma = gamma; p.beta = beta; p.bias1 = bias1; p.bias2 = bias2;
    p.scale1 = scale1; p.scale2 = scale2; p.eps = eps; p.T = T; p.NT = hidden / 32;
    if (C == 192) {
        const size_t lds16 = (size_t)4 * C * 128 + ((size_t)hidden + 3 * C) * 4;
        AL3D_REQUIRE(lds16 <= 160 * 1024, "al3d_tok_mlp_f16x3: hidden = %d does not fit the LDS at C = 192", hidden);
        static bool attr16 = false;
        if (!attr16) {
            if (hipFuncSetAttribute((const void*)tok_mlp16_f16x3_kernel<192>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return al3d_fail(AL3D_ELAUNCH, "al3d_tok_mlp_f16x3: cannot raise the dynamic LDS limit");
            attr16 = true;
        }
        hipLaunchKernelGGL((tok_mlp16_f16x3_kernel<192>), dim3((unsigned)al3d_cdiv(T, 192)), dim3(768), lds16, (hipStream_t)stream, p);
        AL3D_CHECK_LAUNCH("tok_mlp16_f16x3_kernel");
        return AL3D_OK;
    }
    // four waves (128 tokens) on a two-stage ring, two workgroups per CU (default), or eight waves on a three-stage ring, one
    // workgroup per CU (AL3D_TOK_MLP=8x3: tiles requested two ahead; measured the same: the DMA round trip is not the bound)
    static int wide = -1;
    if (wide < 0) { const char* e = getenv("AL3D_TOK_MLP"); wide = e && e[0] == '8'; }
    const int nw = wide ? 8 : 4, ns = wide ? 3 : 2;
    const size_t lds = (size_t)ns * C * 256 + ((size_t)hidden + 3 * C + nw * 32 * 33) * 4;
    const dim3 grid((unsigned)al3d_cdiv(T, nw * 32));
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tok_mlp_f16x3_kernel<96, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)tok_mlp_f16x3_kernel<96, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_tok_mlp_f16x3: cannot raise the dynamic LDS limit");
        attr = true;
    }
    if (wide) hipLaunchKernelGGL((tok_mlp_f16x3_kernel<96, 8, 3>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((tok_mlp_f16x3_kernel<96, 4, 2>), grid, dim3(256), lds, s, p);
    AL3D_CHECK_LAUNCH("tok_mlp_f16x3_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ 7 x 7 window attention, head dim 32
struct TokAttnParams {
    const float* qkv;       // [nwin * 49][3 C]: q | k | v, each [heads][32]
    const float* table;     // [169][heads] relative position bias table
    float* out;             // [nwin * 49][C] f32 or pair rows
    int nwin, C, heads;
    int nwy, nwx;           // windows per image (rows, columns)
    int shift;              // cyclic shift of the block (0: no mask)
    float scale;
    int pair;
    // token-order mode (bias != null): qkv / out rows are the B maps' H x W tokens; the cyclic shift, the padding and the
    // window partition are evaluated from the window's position, a padded position's q / k / v row is the qkv bias
    const float* bias;      // [3 C] or null (window-order mode: rows win * 49 + position)
    int H, W;
};

#define TK_WS 7
#define TK_NT 49

__device__ __forceinline__ int tk_region1(int v, int n, int shift)
{
    return (v >= n - TK_WS ? 1 : 0) + (v >= n - shift ? 1 : 0);
}

// the f16x3 split of 8 values with the packed conversions of sp_split8 (3 instructions per element instead of 5); the
// inputs are pinned first (see tk_split: the high part and the residual must see the same rounded fp32 value)
__device__ __forceinline__ void tk_split8p(float (&v)[8], f16x8& ph, f16x8& pl)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
    uint4 hi, lo;
    sp_split8(v, hi, lo);
    ph = __builtin_bit_cast(f16x8, hi);
    pl = __builtin_bit_cast(f16x8, lo);
}
// 13 y + x of key position min(key, 48) in the 7 x 7 window: the key's part of the relative position index
__host__ __device__ constexpr int tk_kcode(int key) { return (key < TK_NT ? key : TK_NT - 1) + 6 * ((key < TK_NT ? key : TK_NT - 1) / TK_WS); }

// Two waves (= one 128-thread workgroup) per (window, head), one 32-query tile each.  The head's q, k, v rows (49 x 128 B each, 1152+ B apart in
// the qkv matrix) come in by LDS-DMA, eight whole rows per instruction (every 128-byte line fetched once, by one
// instruction); a row's eight 16-byte chunks are stored permuted (chunk q at position q ^ ((row >> 1) & 7), applied
// on the SOURCE side: the LDS side of a DMA is lane-linear) so that the fragment reads are conflict-free.
// C-layout of v_mfma_f32_32x32x16: column = lane & 31, rows in the 16 registers
// (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); S^T puts the KEYS on the rows, so a query's softmax runs down a
// lane's registers (+ one exchange with lane ^ 32), and P^T is already the B operand of O^T = V^T P^T: registers
// 8 s .. 8 s + 7 of a tile are k-step s, in the order key = 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the V^T fragment
// is read in that same order.  Both operands of both products are activations: each is split (xh, xl' = residual
// x 2^11) and the product is  xh yh  +  2^-11 (xh yl' + xl' yh)  with the two brackets in separate accumulators.
#define TK_AROWS 56                   // rows staged per array: 7 DMA instructions of 8 rows; rows 49 .. 55 are zero
#define TK_ABYTES (TK_AROWS * 128)

__device__ __forceinline__ unsigned tk_arow_off(int row, int chunk)      // byte offset of 16-byte chunk `chunk` of a staged row
{
    const int r = row < TK_AROWS ? row : TK_AROWS - 1;                   // rows 56 .. 63 of a tile read a zero row
    return (unsigned)(r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
}

__global__ __launch_bounds__(128, 3) void tok_window_attention_kernel(TokAttnParams p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char stg[3 * TK_ABYTES];    // k | q | v
    __shared__ float tbl[1][176];
    // two waves per (window, head): they share the staged q / k / v rows and take one 32-query tile each; K fragments are
    // read (and split) where they are used instead of being held: <= 168 registers, three waves per SIMD, six workgroups
    // per CU by LDS -- twelve resident waves with half the dependent chain each (one wave per item held 7 per CU)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int item = blockIdx.x;
    const int win = item / p.heads, head = item - win * p.heads;
    const int c = lane & 31, h = lane >> 5;
    const int ld = 3 * p.C;
    const float* base = p.qkv + (int64_t)win * TK_NT * ld + head * 32;
    const unsigned stg_base = (unsigned)(size_t)(tk_lds_void*)stg;
    const int wi = win % (p.nwy * p.nwx), wb = win / (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi - wy * p.nwx;
    // token-order mode: token row of window position `row`, -1 for padding (shifted[hp] = padded[(hp + shift) % Hp])
    auto token_of = [&](int row) __attribute__((always_inline)) -> int {
        const int ty = (row * 37) >> 8, tx = row - ty * TK_WS;
        int hs = wy * TK_WS + ty + p.shift, ws = wx * TK_WS + tx + p.shift;
        hs -= hs >= p.nwy * TK_WS ? p.nwy * TK_WS : 0;
        ws -= ws >= p.nwx * TK_WS ? p.nwx * TK_WS : 0;
        return hs < p.H && ws < p.W ? (wb * p.H + hs) * p.W + ws : -1;
    };
    // the rows first (their latency is the longest: per row group ONE source row address serves the k, q and v pieces), then
    // the position-bias table and the region codes in its shadow -- with the table load in front every item began by
    // waiting for it before a single row was requested (35 % of an item's life by per-phase time stamps)
    {
        const int rl = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int it0 = 0; it0 < 4; ++it0) {
            const int it = 2 * it0 + wave;                      // the row groups of an array alternate between the waves
            if (it >= 7) continue;
            const int row = it * 8 + rl;
            const int chunk = pos ^ ((row >> 1) & 7);
            const bool live = row < TK_NT;
            const float* rp = g_tok_zero;
            if (live) {
                rp = base + (int64_t)row * ld;
                if (p.bias) {
                    const int tok = token_of(row);
                    rp = (tok >= 0 ? p.qkv + (int64_t)tok * ld : p.bias) + head * 32;
                }
            }
            rp += chunk * 4;
#pragma unroll
            for (int arr = 0; arr < 3; ++arr) {
                const int aoff = live ? (arr == 0 ? p.C : arr == 1 ? 0 : 2 * p.C) : 0;
                const unsigned dst = __builtin_amdgcn_readfirstlane(stg_base + arr * TK_ABYTES + it * 1024);
                __builtin_amdgcn_global_load_lds((tk_gbl_void*)(rp + aoff), (tk_lds_void*)(size_t)dst, 16, 0, 0);
            }
        }
    }
    float tv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int t = threadIdx.x + 128 * k; tv[k] = t < 169 ? p.table[t * p.heads + head] : 0.f; }
    // shifted-window regions of the window's 7 rows / 7 columns, two bits each (uniform): tokens attend inside a region
    int rycode = 0, rxcode = 0;
    if (p.shift > 0) {
        for (int t = 0; t < TK_WS; ++t) {
            rycode |= tk_region1(wy * TK_WS + t, p.nwy * TK_WS, p.shift) << (2 * t);
            rxcode |= tk_region1(wx * TK_WS + t, p.nwx * TK_WS, p.shift) << (2 * t);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int t = threadIdx.x + 128 * k; if (t < 176) tbl[0][t] = tv[k]; }
    tk_wait_vm<0>();
    __syncthreads();                                   // both waves' shares of k, q and v have landed
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    const bool masked = p.shift > 0;
    // one 32-query tile at a time (a real loop: the state of a tile -- 64 logit + 32 output accumulators -- is live only
    // inside its iteration, which is what lets several waves share a SIMD)
    for (int j = wave; j <= wave; ++j) {                // this wave's query tile
        const int query = 32 * j + c;
        f16x8 qh[2], ql[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            tk_f32x4 lo, hi;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(lo), "=&v"(hi)
                         : "v"(stg_base + TK_ABYTES + tk_arow_off(query, 4 * s + 2 * h)),
                           "v"(stg_base + TK_ABYTES + tk_arow_off(query, 4 * s + 2 * h + 1))
                         : "memory");
            float qv[8] = {lo[0] * p.scale, lo[1] * p.scale, lo[2] * p.scale, lo[3] * p.scale,
                                 hi[0] * p.scale, hi[1] * p.scale, hi[2] * p.scale, hi[3] * p.scale};
            tk_split8p(qv, qh[s], ql[s]);
        }
        f32x16 sm[2], sc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[i][r] = 0.f; sc[i][r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                tk_f32x4 lo, hi;                                 // K fragment (A operand: rows = keys), split here
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(lo), "=&v"(hi)
                             : "v"(stg_base + tk_arow_off(32 * i + c, 4 * s + 2 * h)), "v"(stg_base + tk_arow_off(32 * i + c, 4 * s + 2 * h + 1))
                             : "memory");
                f16x8 kh, kl;
                { float kv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; tk_split8p(kv, kh, kl); }
                sc[i] = TK_MFMA(kl, qh[s], sc[i]);
                sc[i] = TK_MFMA(kh, ql[s], sc[i]);
                sm[i] = TK_MFMA(kh, qh[s], sm[i]);
            }
        // logits -> probabilities, in place in sm[i] (rows = keys, column = this lane's query)
        const int qq = query < TK_NT ? query : TK_NT - 1;
        const int qy = (qq * 37) >> 8, qx = qq - TK_WS * qy;
        const int qcode = qq + 6 * qy + 84;                          // 13 y + x + 84
        // bit k of `diff`: key k lies in ANOTHER shifted-window region than this query (-100 on its logit).  A bit mask per
        // query instead of a region lookup per element: that form (an LDS read behind `if (masked)`) made hipcc serialise
        // 64 LDS round trips per tile
        unsigned dlo = 0u, dhi = 0u;
        if (masked) {
            const int myry = (rycode >> (2 * qy)) & 3, myrx = (rxcode >> (2 * qx)) & 3;
            unsigned colmask = 0u;
            unsigned long long same = 0ull;
#pragma unroll
            for (int t = 0; t < TK_WS; ++t) colmask |= (unsigned)(((rxcode >> (2 * t)) & 3) == myrx) << t;
#pragma unroll
            for (int t = 0; t < TK_WS; ++t)
                if (((rycode >> (2 * t)) & 3) == myry) same |= (unsigned long long)colmask << (TK_WS * t);
            const unsigned long long diff = ~same >> (4 * h);        // the lane's keys are c + 4 h with compile-time c
            dlo = (unsigned)diff;
            dhi = (unsigned)(diff >> 32);
        }
        const float* tq = tbl[0] + qcode;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float tb[16];                                            // the tile's 16 bias lookups first, then their uses
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cc = 32 * i + (r & 3) + 8 * (r >> 2);      // key = cc + 4 h
                tb[r] = tq[-(h ? tk_kcode(cc + 4) : tk_kcode(cc))];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cc = 32 * i + (r & 3) + 8 * (r >> 2);
                float v = sm[i][r] + sc[i][r] * 0.00048828125f;
                v += tb[r];
                if (masked) v += (float)(((cc < 32 ? dlo : dhi) >> (cc & 31)) & 1u) * -100.0f;
                if (cc + 4 >= TK_NT) v = (cc >= TK_NT || h) ? -INFINITY : v;
                sm[i][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // e^(v - mx) = 2^((v - mx) log2 e): the product in two pieces so that the argument of v_exp_f32 carries no
                // rounding of its own beyond 2^-24 relative (|v - mx| <= ~100 here)
                const float d = sm[i][r] - mx;
                const float t = __builtin_fmaf(d, 1.44269502162933349609f, d * 1.92596299112661746e-8f);
                const float e = __builtin_amdgcn_exp2f(t);
                sm[i][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // O^T[d][query] = sum_key V[key][d] P[query][key]; P is normalised AFTER the product (one multiply per output)
        f32x16 om, oc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { om[r] = 0.f; oc[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float vv[8];
                unsigned va[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = 32 * i + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
                    va[e] = stg_base + 2 * TK_ABYTES + tk_arow_off(key, c >> 2) + ((c & 3) << 2);
                }
                // one block: the wait belongs to the reads (separate asm statements could be scheduled apart from their uses)
                asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                             "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(vv[0]), "=&v"(vv[1]), "=&v"(vv[2]), "=&v"(vv[3]), "=&v"(vv[4]), "=&v"(vv[5]), "=&v"(vv[6]), "=&v"(vv[7])
                             : "v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "v"(va[4]), "v"(va[5]), "v"(va[6]), "v"(va[7])
                             : "memory");
                f16x8 vh, vl, ph, pl;
                tk_split8p(vv, vh, vl);
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = sm[i][8 * s + e];
                tk_split8p(pv, ph, pl);
                oc = TK_MFMA(vl, ph, oc);
                oc = TK_MFMA(vh, pl, oc);
                om = TK_MFMA(vh, ph, om);
            }
        if (query >= TK_NT) continue;
        // rows of O^T are d = (r & 3) + 8 (r >> 2) + 4 h: four consecutive channels per register quad
        int64_t out_row = (int64_t)win * TK_NT + query;
        if (p.bias) {
            out_row = token_of(query);
            if (out_row < 0) continue;                       // a padded position's output is cropped
        }
        float* orow = p.out + out_row * p.C + head * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (om[4 * g + e] + oc[4 * g + e] * 0.00048828125f) * inv;
            if (p.pair) {                                // group g of the head: xh[8] | xl'[8]; this lane owns elements 4 h .. 4 h + 3
                _Float16 hh[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) tk_split(y[e], hh[e], ll[e]);
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 vh4 = {hh[0], hh[1], hh[2], hh[3]}, vl4 = {ll[0], ll[1], ll[2], ll[3]};
                char* o = reinterpret_cast<char*>(orow + 8 * g);
                *reinterpret_cast<uint2*>(o + 8 * h) = __builtin_bit_cast(uint2, vh4);
                *reinterpret_cast<uint2*>(o + 16 + 8 * h) = __builtin_bit_cast(uint2, vl4);
            } else {
                *reinterpret_cast<float4*>(orow + 8 * g + 4 * h) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
}

extern "C" int al3d_tok_window_attention_f32(const float* qkv, const float* table, int nwin, int C, int heads,
                                             int win_rows, int win_cols, int shift, float scale, int out_pair,
                                             float* out, void* stream)
{
    AL3D_REQUIRE(qkv && table && out, "al3d_tok_window_attention_f32: null pointer");
    AL3D_REQUIRE(nwin >= 0 && heads >= 1 && C == heads * 32, "al3d_tok_window_attention_f32: C=%d must be heads (%d) x 32", C, heads);
    AL3D_REQUIRE(win_rows >= 1 && win_cols >= 1 && nwin % (win_rows * win_cols) == 0,
                 "al3d_tok_window_attention_f32: nwin=%d is not a whole number of %d x %d window grids", nwin, win_rows, win_cols);
    AL3D_REQUIRE(shift >= 0 && shift < TK_WS, "al3d_tok_window_attention_f32: shift=%d outside [0, 7)", shift);
    AL3D_REQUIRE((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "al3d_tok_window_attention_f32: qkv / out must be 16-byte aligned");
    if (nwin == 0) return AL3D_OK;
    TokAttnParams p{qkv, table, out, nwin, C, heads, win_rows, win_cols, shift, scale, out_pair, nullptr, 0, 0};
    const int64_t items = (int64_t)nwin * heads;
    hipLaunchKernelGGL(tok_window_attention_kernel, dim3((unsigned)items), dim3(128), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("tok_window_attention_kernel");
    return AL3D_OK;
}

extern "C" int al3d_tok_window_attention_tokens_f32(const float* qkv, const float* bias_qkv, const float* table, int B, int H,
                                                    int W, int C, int heads, int shift, float scale, int out_pair,
                                                    float* out, void* stream)
{
    AL3D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (int64_t)B * H * W < ((int64_t)1 << 31), "al3d_tok_window_attention_tokens_f32: bad map size");
    if (B == 0) return AL3D_OK;
    AL3D_REQUIRE(qkv && bias_qkv && table && out, "al3d_tok_window_attention_tokens_f32: null pointer (a model without qkv bias passes zeros)");
    AL3D_REQUIRE(heads >= 1 && C == heads * 32, "al3d_tok_window_attention_tokens_f32: C=%d must be heads (%d) x 32", C, heads);
    AL3D_REQUIRE(shift >= 0 && shift < TK_WS, "al3d_tok_window_attention_tokens_f32: shift=%d outside [0, 7)", shift);
    AL3D_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)bias_qkv) & 15) == 0, "al3d_tok_window_attention_tokens_f32: qkv / bias / out must be 16-byte aligned");
    const int nwy = (H + TK_WS - 1) / TK_WS, nwx = (W + TK_WS - 1) / TK_WS;
    const int64_t items = (int64_t)B * nwy * nwx * heads;
    AL3D_REQUIRE(items < ((int64_t)1 << 31), "al3d_tok_window_attention_tokens_f32: too many (window, head) items");
    TokAttnParams p{qkv, table, out, B * nwy * nwx, C, heads, nwy, nwx, shift, scale, out_pair, bias_qkv, H, W};
    hipLaunchKernelGGL(tok_window_attention_kernel, dim3((unsigned)items), dim3(128), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("tok_window_attention_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ multi-head attention, head dim 16, any key count
// The TransFusion query decoder (bevfusion/mmdet3d/models/utils/transformer.py:71-112: nn.MultiheadAttention with 8
// heads of 16 channels; 200 queries against themselves, then against the 180 x 180 = 32,400 BEV cells).
// One wave per (sample, head, 32-query tile, key chunk): S^T = K (Q scale)^T per 32-key tile -- head dim 16 is exactly
// one k-step of v_mfma_f32_32x32x16 -- an online softmax down the accumulator registers (running max / sum per query
// = per lane), and O^T += V^T P^T with P taken from the accumulators as the B operand (rows of O^T = the 16 channels;
// the upper half of the 32-row tile is idle).  Both operands split as in the window kernel (main + 2^-11 correction
// accumulators).  Each wave writes (max, sum, O[16]) of its chunk; tok_mha16_combine_kernel merges the chunks.
struct TokMhaParams {
    const float* q;         // [B][Pq][ldq], this head's 16 channels at column head * 16
    const float* k;         // [B][Pk][ldk]
    const float* v;         // [B][Pk][ldv]
    float* part;            // [B][heads][chunks][qtiles * 32][18]: running max, sum, O[16]
    int B, heads, Pq, Pk, ldq, ldk, ldv;
    int qtiles, chunks, keys_per_chunk;          // keys_per_chunk: a multiple of 32
    float scale;
};

__global__ __launch_bounds__(64) void tok_mha16_kernel(TokMhaParams p)
{
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    int id = blockIdx.x;
    const int chunk = id % p.chunks; id /= p.chunks;
    const int qt = id % p.qtiles; id /= p.qtiles;
    const int head = id % p.heads;
    const int b = id / p.heads;
    const int query = qt * 32 + c;
    f16x8 qh, ql;
    {
        float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (query < p.Pq) {
            const float* qp = p.q + ((int64_t)b * p.Pq + query) * p.ldq + head * 16 + 8 * h;
            const float4 a = *reinterpret_cast<const float4*>(qp), b4 = *reinterpret_cast<const float4*>(qp + 4);
            qv[0] = a.x * p.scale; qv[1] = a.y * p.scale; qv[2] = a.z * p.scale; qv[3] = a.w * p.scale;
            qv[4] = b4.x * p.scale; qv[5] = b4.y * p.scale; qv[6] = b4.z * p.scale; qv[7] = b4.w * p.scale;
        }
        tk_split8(qv, qh, ql);
    }
    const int key0 = chunk * p.keys_per_chunk;
    const int key1 = key0 + p.keys_per_chunk < p.Pk ? key0 + p.keys_per_chunk : p.Pk;
    const float* kb = p.k + (int64_t)b * p.Pk * p.ldk + head * 16;
    const float* vb = p.v + (int64_t)b * p.Pk * p.ldv + head * 16;
    float run_max = -INFINITY, run_sum = 0.f;
    f32x16 om, oc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { om[r] = 0.f; oc[r] = 0.f; }
    for (int kt = key0; kt < key1; kt += 32) {
        // K tile: A operand, lane (key c, half h) holds K[key][8 h .. 8 h + 7]
        f16x8 kh, kl;
        {
            float kv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (kt + c < key1) {
                const float* kp = kb + (int64_t)(kt + c) * p.ldk + 8 * h;
                const float4 a = *reinterpret_cast<const float4*>(kp), b4 = *reinterpret_cast<const float4*>(kp + 4);
                kv[0] = a.x; kv[1] = a.y; kv[2] = a.z; kv[3] = a.w; kv[4] = b4.x; kv[5] = b4.y; kv[6] = b4.z; kv[7] = b4.w;
            }
            tk_split8(kv, kh, kl);
        }
        // V^T fragments of the tile's two k-steps (issued early: their latency hides behind the logits)
        float vv[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = kt + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
                vv[s][e] = (c < 16 && key < key1) ? vb[(int64_t)key * p.ldv + c] : 0.f;
            }
        f32x16 sm, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm[r] = 0.f; sc[r] = 0.f; }
        sc = TK_MFMA(kl, qh, sc);
        sc = TK_MFMA(kh, ql, sc);
        sm = TK_MFMA(kh, qh, sm);
        float mx = run_max;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = key < key1 ? sm[r] + sc[r] * 0.00048828125f : -INFINITY;
            sm[r] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));              // every tile holds at least one real key: mx is finite
        const float resc = __builtin_amdgcn_exp2f((run_max - mx) * 1.44269504088896340736f);     // 0 on the first tile
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = sm[r] - mx;
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(d, 1.44269502162933349609f, d * 1.92596299112661746e-8f));
            sm[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32);
        run_sum = run_sum * resc + sum;
        run_max = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { om[r] *= resc; oc[r] *= resc; }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 vh, vl, ph, pl;
            tk_split8(vv[s], vh, vl);
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = sm[8 * s + e];
            tk_split8(pv, ph, pl);
            oc = TK_MFMA(vl, ph, oc);
            oc = TK_MFMA(vh, pl, oc);
            om = TK_MFMA(vh, ph, om);
        }
    }
    // rows of O^T: d = (r & 3) + 8 (r >> 2) + 4 h; d < 16 <=> r < 8
    float* o = p.part + ((((int64_t)b * p.heads + head) * p.chunks + chunk) * (p.qtiles * 32) + query) * 18;
    if (h == 0) { o[0] = run_max; o[1] = run_sum; }
#pragma unroll
    for (int r = 0; r < 8; ++r) o[2 + (r & 3) + 8 * (r >> 2) + 4 * h] = om[r] + oc[r] * 0.00048828125f;
}

// out[b][query][head * 16 + d] = sum_c e^(m_c - M) O_c[d] / sum_c e^(m_c - M) l_c
__global__ __launch_bounds__(256) void tok_mha16_combine_kernel(const float* __restrict__ part, int B, int heads, int chunks,
                                                                int qrows, int Pq, float* __restrict__ out, int ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * heads * Pq * 16) return;
    const int d = (int)(t & 15);
    int64_t r = t >> 4;
    const int query = (int)(r % Pq); r /= Pq;
    const int head = (int)(r % heads);
    const int b = (int)(r / heads);
    const float* base = part + (((int64_t)b * heads + head) * chunks * qrows + query) * 18;
    float M = -INFINITY;
    for (int c = 0; c < chunks; ++c) M = fmaxf(M, base[(int64_t)c * qrows * 18]);
    float num = 0.f, den = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float* q = base + (int64_t)c * qrows * 18;
        const float w = expf(q[0] - M);
        num += w * q[2 + d];
        den += w * q[1];
    }
    out[((int64_t)b * Pq + query) * ldo + head * 16 + d] = num / den;
}

extern "C" int64_t al3d_tok_mha16_workspace_bytes(int B, int heads, int Pq, int Pk)
{
    if (B < 1 || heads < 1 || Pq < 1 || Pk < 1) return 0;
    const int qtiles = (Pq + 31) / 32;
    int chunks = (Pk + 1023) / 1024;
    return al3d_align((int64_t)B * heads * chunks * qtiles * 32 * 18 * 4, 256);
}

extern "C" int al3d_tok_mha16_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int B, int heads,
                                  int Pq, int Pk, float scale, float* out, int ldo, void* workspace, void* stream)
{
    AL3D_REQUIRE(q && k && v && out && workspace, "al3d_tok_mha16_f32: null pointer");
    AL3D_REQUIRE(B >= 1 && heads >= 1 && Pq >= 1 && Pk >= 1, "al3d_tok_mha16_f32: bad shape");
    AL3D_REQUIRE(ldq >= heads * 16 && ldk >= heads * 16 && ldv >= heads * 16 && ldo >= heads * 16 && ldq % 4 == 0 && ldk % 4 == 0,
                 "al3d_tok_mha16_f32: row pitches must cover heads x 16 channels (q, k pitches multiples of 4)");
    AL3D_REQUIRE((((uintptr_t)q | (uintptr_t)k) & 15) == 0, "al3d_tok_mha16_f32: q / k must be 16-byte aligned");
    TokMhaParams p;
    p.q = q; p.k = k; p.v = v; p.part = (float*)workspace;
    p.B = B; p.heads = heads; p.Pq = Pq; p.Pk = Pk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
    p.qtiles = (Pq + 31) / 32;
    p.chunks = (Pk + 1023) / 1024;
    p.keys_per_chunk = (int)al3d_align(al3d_cdiv(Pk, p.chunks), 32);
    p.chunks = (int)al3d_cdiv(Pk, p.keys_per_chunk);       // no empty chunk: every partial holds at least one key
    p.scale = scale;
    const int64_t waves = (int64_t)B * heads * p.qtiles * p.chunks;
    AL3D_REQUIRE(waves < ((int64_t)1 << 31), "al3d_tok_mha16_f32: too many work items");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tok_mha16_kernel, dim3((unsigned)waves), dim3(64), 0, s, p);
    AL3D_CHECK_LAUNCH("tok_mha16_kernel");
    const int64_t n = (int64_t)B * heads * Pq * 16;
    hipLaunchKernelGGL(tok_mha16_combine_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0, s, (const float*)workspace, B,
                       heads, p.chunks, p.qtiles * 32, Pq, out, ldo);
    AL3D_CHECK_LAUNCH("tok_mha16_combine_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ fused attention half of a Swin block
// x += proj(W-MSA(LN1(x))) as ONE kernel (C = 96 / 192, the two stages whose separate LN1 / qkv / attention / proj
// launches are bound by the 44 C bytes per token they move; this kernel reads the window's rows once and writes them once:
// 8 C).  Per 7 x 7 window, two waves per head:
//   1. all waves: LayerNorm of the window's 49 rows, gathered arithmetically (cyclic shift + window partition + padding
//      from the window's position: no row map) -> pair rows (xh, xl') in LDS; padding and rows 49 .. 63 are zero rows;
//   2. wave (head, 0) forms K^T = Wk^T xn^T (weights = A operand; tokens across the lanes, head channels down the
//      accumulator registers), wave (head, 1) forms V = xn Wv (head channel = lane, tokens down the registers), both token
//      tiles each; in these orientations the accumulators ARE MFMA A fragments after bias + split (K for S^T = K Q^T, V^T
//      for O^T = V^T P^T with the keys in the order P^T's rows leave the softmax), written to LDS lane-linear: 1 KB per
//      (tile, channel step, plane), read back conflict-free.  Both waves form Q^T of their own query tile (registers).
//      q, k, v never exist in memory;
//   3. each wave: S^T, softmax down the registers (relative position bias from the table, shifted-window regions as a bit
//      mask per query: no LDS, no branch per element), O^T of its query tile -> pair rows in LDS (over the normalised
//      rows, dead after the barrier that publishes K and V);
//   4. each wave: output channels 32 head .. + 31 of Wp^T O^T for its token tile, + bias + residual, written back to the
//      rows the window came from.
// The weights never touch LDS: each wave streams its fragments (image per head: parts k | v | q | proj, each
// [C/16][2 planes][64 lanes][8], MFMA lane order) from L2 into a register ring AB_D2 steps ahead; the first steps are
// requested before the LayerNorm's rows.  160 registers: three waves per SIMD.  A workgroup's waves are dealt to the four
// SIMDs round-robin from SIMD 0, so a workgroup holds a multiple of four waves: twelve = one window at C = 192, TWO windows
// at C = 96 (WPB).  f16x3 arithmetic as everywhere: weights x activations = three products into one accumulator,
// activations x activations = main + 2^-11 correction accumulators.
// (A first form -- one wave per head, K / V / both query tiles in registers, 256 registers, 1.5 waves per SIMD -- measured
// 0.90 / 0.64 ms per launch at C = 96 / 192 against 0.80 / 0.51 for this one)
struct TokAttnBlockParams {
    float* x;                // [B * H * W][C] residual stream, updated in place
    const _Float16* image;
    const float* gamma;      // LN1 [C]
    const float* beta;
    const float* bias_qkv;   // [3 C]: q | k | v
    const float* bias_proj;  // [C]
    const float* table;      // [169][heads]
    float eps, scale_qkv, scale_proj, scale;
    int B, H, W, nwy, nwx, shift;
};

#define AB_D2 3              // ring depth of the two-wave form (registers: three waves per SIMD)

template <int C, int WPB>
__global__ __launch_bounds__(C * 4 * WPB, 3) void tok_attn_block_f16x3_kernel(TokAttnBlockParams p)
{
    constexpr int NH = C / 32, NWV = 2 * NH, KC = C / 16, NTHR = NWV * 64, PITCH = C * 4 + 16;
    constexpr int LDSW = 64 * PITCH + NH * 16384 + NH * 176 * 4 + NWV * 128 * 4;       // bytes of LDS per window
    constexpr int NSTEP = 3 * KC;                         // this wave's stream: k or v | q | proj
    static_assert(KC % AB_D2 == 0, "the ring holds a whole number of steps per part");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ab_smem[];
    const int lw = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR);          // window of this workgroup
    unsigned char* xn = ab_smem + lw * LDSW;              // 64 pair rows; after barrier 2: the attention output
    unsigned char* kb = xn + 64 * PITCH;                  // [head][tile][step s][plane][64 lanes][16 B]
    unsigned char* vb = kb + NH * 8192;
    const int tid = threadIdx.x - lw * NTHR, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = wave >> 1, role = wave & 1;            // role = 0: forms K; 1: forms V; also this wave's query / token tile
    float* tbl = reinterpret_cast<float*>(vb + NH * 8192) + hd * 176;
    float* bsm = reinterpret_cast<float*>(vb + NH * 8192 + NH * 176 * 4) + wave * 128;      // k | q | proj | v rows of this head
    const int nwin = p.B * p.nwy * p.nwx;
    const bool live = (int)blockIdx.x * WPB + lw < nwin;   // the last workgroup of an odd window count runs one window dry
    const int win = live ? blockIdx.x * WPB + lw : nwin - 1;
    const int wi = win % (p.nwy * p.nwx), b = win / (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi - wy * p.nwx;
    auto token_of = [&](int row) __attribute__((always_inline)) -> int {
        const int ty = row / TK_WS, tx = row - ty * TK_WS;
        int hs = wy * TK_WS + ty + p.shift, ws = wx * TK_WS + tx + p.shift;
        hs -= hs >= p.nwy * TK_WS ? p.nwy * TK_WS : 0;
        ws -= ws >= p.nwx * TK_WS ? p.nwx * TK_WS : 0;
        return live && row < TK_NT && hs < p.H && ws < p.W ? (b * p.H + hs) * p.W + ws : -1;
    };

    // ---- the weight stream of this wave: steps [0, KC) = k or v part, [KC, 2 KC) = q, [2 KC, 3 KC) = proj
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.image) + (size_t)hd * (4 * KC * 2048) + lane * 16;
    f16x8 rwh[AB_D2], rwl[AB_D2];
    auto issue = [&](int g) __attribute__((always_inline)) {
        if (g < NSTEP) {
            const int f = g < KC ? role * KC + g : g + KC;           // image parts: k | v | q | proj
            rwh[g % AB_D2] = *reinterpret_cast<const f16x8*>(wimg + f * 2048);
            rwl[g % AB_D2] = *reinterpret_cast<const f16x8*>(wimg + f * 2048 + 1024);
        }
    };
#pragma unroll
    for (int g = 0; g < AB_D2; ++g) issue(g);

    // the position-bias table and this wave's bias rows are REQUESTED here and stored to LDS after the LayerNorm's rows have
    // been requested: a load -> LDS copy in front would make every wave wait one memory round trip before it asks for its rows
    float tv[3], bv[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) tv[k] = lane + 64 * k < 169 ? p.table[(lane + 64 * k) * NH + hd] : 0.f;
    {
        const int part = lane >> 5;                        // lanes 0-31: k, proj; 32-63: q, v
        bv[0] = p.bias_qkv[(part == 0 ? C : 0) + 32 * hd + fr];
        bv[1] = part == 0 ? p.bias_proj[32 * hd + fr] : p.bias_qkv[2 * C + 32 * hd + fr];
    }
    int rycode = 0, rxcode = 0;
    if (p.shift > 0)
        for (int t = 0; t < TK_WS; ++t) {
            rycode |= tk_region1(wy * TK_WS + t, p.nwy * TK_WS, p.shift) << (2 * t);
            rxcode |= tk_region1(wx * TK_WS + t, p.nwx * TK_WS, p.shift) << (2 * t);
        }
    // ---- 1. LayerNorm: C / 24 lanes per row (4 or 8), three groups of 8 channels per lane, every row of the window in one
    // pass (the per-row work -- token index, divisions, root -- is then done by 4 or 8 lanes, not by 16 for three passes);
    // rows 49 .. 63 and padding positions come out as zero rows: keys with zero weight, finite values
    {
        constexpr int LPR = C / 24;
        static_assert(NTHR / LPR >= 64, "one pass covers the 64 rows of the tile");
        const int sub = tid % LPR, row = tid / LPR;
        const int src = token_of(row);
        float v[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int g = sub + LPR * t;
            const float* s = p.x + (int64_t)(src < 0 ? 0 : src) * C + 8 * g;          // unconditional (clamped) loads
            const float4 a = *reinterpret_cast<const float4*>(s), b4 = *reinterpret_cast<const float4*>(s + 4);
            v[t][0] = a.x; v[t][1] = a.y; v[t][2] = a.z; v[t][3] = a.w; v[t][4] = b4.x; v[t][5] = b4.y; v[t][6] = b4.z; v[t][7] = b4.w;
        }
        float ga[3][8], be[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int g = sub + LPR * t;
            const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + 8 * g), g1 = *reinterpret_cast<const float4*>(p.gamma + 8 * g + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.beta + 8 * g), b1 = *reinterpret_cast<const float4*>(p.beta + 8 * g + 4);
            ga[t][0] = g0.x; ga[t][1] = g0.y; ga[t][2] = g0.z; ga[t][3] = g0.w; ga[t][4] = g1.x; ga[t][5] = g1.y; ga[t][6] = g1.z; ga[t][7] = g1.w;
            be[t][0] = b0.x; be[t][1] = b0.y; be[t][2] = b0.z; be[t][3] = b0.w; be[t][4] = b1.x; be[t][5] = b1.y; be[t][6] = b1.z; be[t][7] = b1.w;
        }
        if (role == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) if (lane + 64 * k < 176) tbl[lane + 64 * k] = tv[k];
        }
        bsm[lane] = bv[0];
        bsm[64 + lane] = bv[1];
        // sum over the row's LPR adjacent lanes: quad exchanges, then (8 lanes) the mirrored half row, whose lanes all hold
        // their quad's sum by then
        auto rowsum = [](float x) __attribute__((always_inline)) -> float {
#define TK_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
            x += TK_DPP(0xb1);                               // quad_perm [1, 0, 3, 2]
            x += TK_DPP(0x4e);                               // quad_perm [2, 3, 0, 1]
            if (LPR == 8) x += TK_DPP(0x141);                // row_half_mirror
#undef TK_DPP
            return x;
        };
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[t][e];
        const float mean = rowsum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[t][e] - mean; sq += d * d; }
        const float rstd = 1.0f / sqrtf(rowsum(sq) / (float)C + p.eps);
        if (row < 64) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int g = sub + LPR * t;
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = src < 0 ? 0.f : ((v[t][e] - mean) * rstd) * ga[t][e] + be[t][e];
                uint4 hi, lo;
                sp_split8(y, hi, lo);
                uint4* o = reinterpret_cast<uint4*>(__builtin_assume_aligned(xn + row * PITCH + g * 32, 16));
                o[0] = hi;
                o[1] = lo;
            }
        }
    }
    __syncthreads();

    auto xfrag = [&](int t, int kc, f16x8& xh, f16x8& xl) __attribute__((always_inline)) {
        const unsigned char* s = xn + (32 * t + fr) * PITCH + (2 * kc + fh) * 32;
        xh = *reinterpret_cast<const f16x8*>(s);
        xl = *reinterpret_cast<const f16x8*>(s + 16);
    };
    auto zero16 = [](f32x16& a) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    auto bias16 = [&](const float* bsrc, float (&o)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bsrc + 8 * g + 4 * fh);
            o[4 * g] = b4.x; o[4 * g + 1] = b4.y; o[4 * g + 2] = b4.z; o[4 * g + 3] = b4.w;
        }
    };
    // ---- 2a. K (role 0) or V (role 1) of both token tiles -> LDS as A fragments
    {
        f32x16 a0, a1;
        zero16(a0); zero16(a1);
        auto kv_loop = [&](auto wa) __attribute__((always_inline)) {      // wa: the weights are the A operand (K^T = Wk^T xn^T); else V = xn Wv
            constexpr bool WA = decltype(wa)::value;
            f16x8 fx[2][4];
            xfrag(0, 0, fx[0][0], fx[0][1]);
            xfrag(1, 0, fx[0][2], fx[0][3]);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const f16x8 wh = rwh[kc % AB_D2], wl = rwl[kc % AB_D2];
                const f16x8 wd = tk_lift_down(wh);
                if (kc + 1 < KC) {
                    xfrag(0, kc + 1, fx[(kc + 1) & 1][0], fx[(kc + 1) & 1][1]);
                    xfrag(1, kc + 1, fx[(kc + 1) & 1][2], fx[(kc + 1) & 1][3]);
                }
                const f16x8 xh0 = fx[kc & 1][0], xl0 = fx[kc & 1][1], xh1 = fx[kc & 1][2], xl1 = fx[kc & 1][3];
                if constexpr (WA) {
                    a0 = TK_MFMA(wd, xl0, a0); a1 = TK_MFMA(wd, xl1, a1);
                    a0 = TK_MFMA(wl, xh0, a0); a1 = TK_MFMA(wl, xh1, a1);
                    a0 = TK_MFMA(wh, xh0, a0); a1 = TK_MFMA(wh, xh1, a1);
                } else {
                    a0 = TK_MFMA(xl0, wd, a0); a1 = TK_MFMA(xl1, wd, a1);
                    a0 = TK_MFMA(xh0, wl, a0); a1 = TK_MFMA(xh1, wl, a1);
                    a0 = TK_MFMA(xh0, wh, a0); a1 = TK_MFMA(xh1, wh, a1);
                }
                issue(kc + AB_D2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (role == 0) kv_loop(std::true_type{});
        else kv_loop(std::false_type{});
        float bk[16];
        bias16(bsm, bk);
        const float bv = bsm[96 + fr];
        unsigned char* dstb = (role == 0 ? kb : vb) + hd * 8192 + lane * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x16& a = t == 0 ? a0 : a1;
                float vv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = a[8 * s + e] * p.scale_qkv + (role == 0 ? bk[8 * s + e] : bv);
                f16x8 oh, ol;
                tk_split8p(vv, oh, ol);
                *reinterpret_cast<f16x8*>(dstb + ((t * 2 + s) * 2) * 1024) = oh;
                *reinterpret_cast<f16x8*>(dstb + ((t * 2 + s) * 2 + 1) * 1024) = ol;
            }
    }
    // ---- 2b. Q^T of this wave's query tile
    f16x8 qh[2], ql[2];
    {
        f32x16 aq;
        zero16(aq);
        f16x8 fx[2][2];
        xfrag(role, 0, fx[0][0], fx[0][1]);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int g = KC + kc;
            const f16x8 wh = rwh[g % AB_D2], wl = rwl[g % AB_D2];
            const f16x8 wd = tk_lift_down(wh);
            if (kc + 1 < KC) xfrag(role, kc + 1, fx[(kc + 1) & 1][0], fx[(kc + 1) & 1][1]);
            aq = TK_MFMA(wd, fx[kc & 1][1], aq);
            aq = TK_MFMA(wl, fx[kc & 1][0], aq);
            aq = TK_MFMA(wh, fx[kc & 1][0], aq);
            if (g + AB_D2 < 2 * KC) issue(g + AB_D2);          // the projection's fragments are requested after the attention
            __builtin_amdgcn_sched_barrier(0);
        }
        float bq[16];
        bias16(bsm + 32, bq);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = (aq[8 * s + e] * p.scale_qkv + bq[8 * s + e]) * p.scale;
            tk_split8p(vv, qh[s], ql[s]);
        }
    }
    __syncthreads();                                         // K and V are in LDS; the normalised rows are dead

    // ---- 3. S^T, softmax, O^T of query tile `role`
    {
        const bool masked = p.shift > 0;
        const int query = 32 * role + fr;
        f32x16 sm[2], sc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { zero16(sm[i]); zero16(sc[i]); }
        const unsigned char* kf = kb + hd * 8192 + lane * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kf + ((i * 2 + s) * 2) * 1024);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(kf + ((i * 2 + s) * 2 + 1) * 1024);
                sc[i] = TK_MFMA(kl, qh[s], sc[i]);
                sc[i] = TK_MFMA(kh, ql[s], sc[i]);
                sm[i] = TK_MFMA(kh, qh[s], sm[i]);
            }
        const int qq = query < TK_NT ? query : TK_NT - 1;
        const int qy = (qq * 37) >> 8, qx = qq - TK_WS * qy;
        const int qcode = qq + 6 * qy + 84;                          // 13 y + x + 84
        // bit k of `diff`: key k lies in ANOTHER shifted-window region than this query (-100 on its logit)
        unsigned dlo = 0u, dhi = 0u;
        if (masked) {
            const int myry = (rycode >> (2 * qy)) & 3, myrx = (rxcode >> (2 * qx)) & 3;
            unsigned colmask = 0u;
            unsigned long long same = 0ull;
#pragma unroll
            for (int t = 0; t < TK_WS; ++t) colmask |= (unsigned)(((rxcode >> (2 * t)) & 3) == myrx) << t;
#pragma unroll
            for (int t = 0; t < TK_WS; ++t)
                if (((rycode >> (2 * t)) & 3) == myry) same |= (unsigned long long)colmask << (TK_WS * t);
            const unsigned long long diff = ~same >> (4 * fh);       // the lane's keys are c + 4 fh with compile-time c
            dlo = (unsigned)diff;
            dhi = (unsigned)(diff >> 32);
        }
        const float* tq = tbl + qcode;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float tb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * i + (r & 3) + 8 * (r >> 2);       // key = c + 4 fh
                tb[r] = tq[-(fh ? tk_kcode(c + 4) : tk_kcode(c))];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * i + (r & 3) + 8 * (r >> 2);
                float v = sm[i][r] + sc[i][r] * 0.00048828125f;
                v += tb[r];
                if (masked) v += (float)(((c < 32 ? dlo : dhi) >> (c & 31)) & 1u) * -100.0f;
                if (c + 4 >= TK_NT) v = (c >= TK_NT || fh) ? -INFINITY : v;
                sm[i][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = sm[i][r] - mx;
                const float t = __builtin_fmaf(d, 1.44269502162933349609f, d * 1.92596299112661746e-8f);
                const float e = __builtin_amdgcn_exp2f(t);
                sm[i][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x16 om, oc;
        zero16(om); zero16(oc);
        const unsigned char* vf = vb + hd * 8192 + lane * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 vh = *reinterpret_cast<const f16x8*>(vf + ((i * 2 + s) * 2) * 1024);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(vf + ((i * 2 + s) * 2 + 1) * 1024);
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = sm[i][8 * s + e];
                f16x8 ph, pl;
                tk_split8p(pv, ph, pl);
                oc = TK_MFMA(vl, ph, oc);
                oc = TK_MFMA(vh, pl, oc);
                om = TK_MFMA(vh, ph, om);
            }
        unsigned char* orow = xn + query * PITCH + (4 * hd) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            _Float16 hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) tk_split((om[4 * g + e] + oc[4 * g + e] * 0.00048828125f) * inv, hh[e], ll[e]);
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 vh4 = {hh[0], hh[1], hh[2], hh[3]}, vl4 = {ll[0], ll[1], ll[2], ll[3]};
            *reinterpret_cast<uint2*>(orow + g * 32 + 8 * fh) = __builtin_bit_cast(uint2, vh4);
            *reinterpret_cast<uint2*>(orow + g * 32 + 16 + 8 * fh) = __builtin_bit_cast(uint2, vl4);
        }
    }
#pragma unroll
    for (int g = 2 * KC; g < 2 * KC + AB_D2; ++g) issue(g);
    const int dst = token_of(32 * role + fr);
    float4 rs[4];
    {
        const float* xr = p.x + (int64_t)(dst < 0 ? 0 : dst) * C + 32 * hd + 4 * fh;
#pragma unroll
        for (int g = 0; g < 4; ++g) rs[g] = *reinterpret_cast<const float4*>(xr + 8 * g);
    }
    __syncthreads();

    // ---- 4. output channels 32 hd .. + 31 of the projection for token tile `role`
    {
        f32x16 a;
        zero16(a);
        f16x8 fx[2][2];
        xfrag(role, 0, fx[0][0], fx[0][1]);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int g = 2 * KC + kc;
            const f16x8 wh = rwh[g % AB_D2], wl = rwl[g % AB_D2];
            const f16x8 wd = tk_lift_down(wh);
            if (kc + 1 < KC) xfrag(role, kc + 1, fx[(kc + 1) & 1][0], fx[(kc + 1) & 1][1]);
            a = TK_MFMA(wd, fx[kc & 1][1], a);
            a = TK_MFMA(wl, fx[kc & 1][0], a);
            a = TK_MFMA(wh, fx[kc & 1][0], a);
            issue(g + AB_D2);
            __builtin_amdgcn_sched_barrier(0);
        }
        float bp[16];
        bias16(bsm + 64, bp);
        if (dst >= 0) {
            float* xr = p.x + (int64_t)dst * C + 32 * hd + 4 * fh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 r4 = rs[g];
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = a[4 * g + e] * p.scale_proj + bp[4 * g + e];
                r4.x += y[0]; r4.y += y[1]; r4.z += y[2]; r4.w += y[3];
                *reinterpret_cast<float4*>(xr + 8 * g) = r4;
            }
        }
    }
}

extern "C" int64_t al3d_tok_attn_block_image_bytes(int C) { return (int64_t)(C / 32) * 4 * (C / 16) * 2048; }

extern "C" int al3d_tok_attn_block_f16x3(float* x, int B, int H, int W, int C, int shift, const float* gamma, const float* beta,
                                         float eps, const void* image, float scale_qkv, const float* bias_qkv,
                                         float scale_proj, const float* bias_proj, const float* table, float attn_scale,
                                         void* stream)
{
    AL3D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (int64_t)B * H * W < ((int64_t)1 << 31), "al3d_tok_attn_block_f16x3: bad map size");
    if (B == 0) return AL3D_OK;
    AL3D_REQUIRE(x && gamma && beta && image && bias_qkv && bias_proj && table, "al3d_tok_attn_block_f16x3: null pointer");
    AL3D_REQUIRE(C == 96 || C == 192, "al3d_tok_attn_block_f16x3: built for C = 96 and 192 (the bandwidth-bound stages), got %d", C);
    AL3D_REQUIRE(shift >= 0 && shift < TK_WS, "al3d_tok_attn_block_f16x3: shift=%d outside [0, 7)", shift);
    AL3D_REQUIRE((((uintptr_t)x | (uintptr_t)image | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias_qkv | (uintptr_t)bias_proj) & 15) == 0,
                 "al3d_tok_attn_block_f16x3: x / image / gamma / beta / biases must be 16-byte aligned");
    const int nwy = (H + TK_WS - 1) / TK_WS, nwx = (W + TK_WS - 1) / TK_WS;
    AL3D_REQUIRE((int64_t)B * nwy * nwx < ((int64_t)1 << 31), "al3d_tok_attn_block_f16x3: too many windows");
    TokAttnBlockParams p{x, (const _Float16*)image, gamma, beta, bias_qkv, bias_proj, table, eps, scale_qkv, scale_proj,
                         attn_scale, B, H, W, nwy, nwx, shift};
    const int NH = C / 32;
    const size_t lds = (size_t)64 * (C * 4 + 16) + (size_t)NH * 16384 + (size_t)NH * 176 * 4 + (size_t)NH * 2 * 128 * 4;   // per window
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tok_attn_block_f16x3_kernel<96, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)tok_attn_block_f16x3_kernel<192, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_tok_attn_block_f16x3: cannot raise the dynamic LDS limit");
        attr = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const unsigned nwin = (unsigned)(B * nwy * nwx);
    if (C == 96) hipLaunchKernelGGL((tok_attn_block_f16x3_kernel<96, 2>), dim3((nwin + 1) / 2), dim3(768), 2 * lds, s, p);
    else hipLaunchKernelGGL((tok_attn_block_f16x3_kernel<192, 1>), dim3(nwin), dim3(768), lds, s, p);
    AL3D_CHECK_LAUNCH("tok_attn_block_f16x3_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ fused patch embedding (embed dim 96)
// mmdet PatchEmbed = Conv2d(3, 96, 4, stride 4) + LayerNorm(96) as ONE kernel: the three launches it replaces (patch rows,
// token GEMM with K = 48, LayerNorm) read and write the 415 MB token matrix (16 samples) two and a half times; this one
// reads the image once (208 MB) and writes the tokens once.  A wave owns 32 tokens at a time: the 48 values of a token's
// 4 x 4 x 3 patch are its B-operand rows (k = (ky * 4 + kx) * 3 + c: every aligned group of four k is 16 contiguous bytes
// of one image row -- W % 4 == 0 keeps a patch inside the image horizontally), split in registers; out^T = W P^T with the
// weights as A operand (nine fragments per plane, 18 KB in LDS, lane order); the accumulators
// hold a token's 96 channels in the lane pair (l, l ^ 32): bias, LayerNorm (two-pass, one exchange per moment), and the rows
// leave through the wave's LDS transposition as 16-byte pieces of 128-byte segments.  f16x3 arithmetic.
struct TokPatchEmbedParams {
    const float* img;        // [B][H][W][3] channels-last
    const _Float16* image;   // [3 tiles u][3 steps s][2 planes][64 lanes][8]: projection.weight[32u + lane%32][16s + 8(lane/32) + e]
    const float* bias;       // [96]
    const float* gamma;      // [96] or null: no LayerNorm
    const float* beta;
    float* out;              // [B * TH * TW][96]
    float scale, eps;
    int B, H, W, TH, TW, gpw;
    int64_t T;
};

__global__ __launch_bounds__(256, 2) void tok_patch_embed_f16x3_kernel(TokPatchEmbedParams p)
{
    __shared__ float par[3][96];                          // bias | gamma | beta
    __shared__ float scr_all[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* scr = scr_all[wave];
    for (int i = tid; i < 96; i += 256) {
        par[0][i] = p.bias ? p.bias[i] : 0.f;
        par[1][i] = p.gamma ? p.gamma[i] : 1.f;
        par[2][i] = p.gamma ? p.beta[i] : 0.f;
    }
    __shared__ __attribute__((aligned(16))) unsigned char wsm[18 * 1024];    // the nine fragments x two planes, lane order
    for (int i = tid; i < 18 * 64; i += 256)
        *reinterpret_cast<uint4*>(wsm + i * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.image) + i * 16);
    const unsigned char* wf = wsm + lane * 16;
    __syncthreads();
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * p.gpw;
    for (int gi = 0; gi < p.gpw; ++gi) {
        const int64_t tok0 = (g0 + gi) * 32;
        if (tok0 >= p.T) break;                              // uniform per wave
        const int64_t tok = tok0 + fr;
        const bool live = tok < p.T;
        const int64_t tc = live ? tok : 0;
        const int tx = (int)(tc % p.TW), ty = (int)((tc / p.TW) % p.TH), b = (int)(tc / ((int64_t)p.TW * p.TH));
        // the lane's three 8-value chunks of the patch: k = 16 s + 8 fh + e; four consecutive k = 16 bytes of one image row
        f16x8 ph[3], pl[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float v[8];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int k0 = 16 * s + 8 * fh + 4 * hlf;       // a multiple of 4: inside one (ky) segment of 12
                const int ky = k0 / 12, off = k0 - 12 * ky;
                const int y = 4 * ty + ky;
                const bool ok = live && y < p.H;
                const float4 q = *reinterpret_cast<const float4*>(p.img + (((int64_t)b * p.H + (ok ? y : 0)) * p.W + 4 * tx) * 3 + off);
                v[4 * hlf] = ok ? q.x : 0.f; v[4 * hlf + 1] = ok ? q.y : 0.f; v[4 * hlf + 2] = ok ? q.z : 0.f; v[4 * hlf + 3] = ok ? q.w : 0.f;
            }
            tk_split8p(v, ph[s], pl[s]);
        }
        f32x16 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(wf + ((u * 3 + s) * 2) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(wf + ((u * 3 + s) * 2 + 1) * 1024);
                acc[u] = TK_MFMA(tk_lift_down(wh), pl[s], acc[u]);
                acc[u] = TK_MFMA(wl, ph[s], acc[u]);
                acc[u] = TK_MFMA(wh, ph[s], acc[u]);
            }
        }
        // channel of register r of tile u: 32 u + (r & 3) + 8 (r >> 2) + 4 fh; the other 48 channels of the token sit in lane ^ 32
        float y[3][16];
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(&par[0][32 * u + 8 * g + 4 * fh]);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[u][4 * g + e] = acc[u][4 * g + e] * p.scale + bb[e]; sum += y[u][4 * g + e]; }
            }
        if (p.gamma) {
            sum += __shfl_xor(sum, 32);
            const float mean = sum / 96.0f;
            float sq = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = y[u][r] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 32);
            const float rstd = 1.0f / sqrtf(sq / 96.0f + p.eps);
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 g4 = *reinterpret_cast<const float4*>(&par[1][32 * u + 8 * g + 4 * fh]);
                    const float4 b4 = *reinterpret_cast<const float4*>(&par[2][32 * u + 8 * g + 4 * fh]);
                    const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[u][4 * g + e] = ((y[u][4 * g + e] - mean) * rstd) * ga[e] + be[e];
                }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[fr * 33 + (r & 3) + 8 * (r >> 2) + 4 * fh] = y[u][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane + 64 * q, tk = i >> 3, g = i & 7;        // token tk, channels 4 g .. 4 g + 3 of the tile
                if (tok0 + tk < p.T)
                    *reinterpret_cast<float4*>(p.out + (tok0 + tk) * 96 + 32 * u + 4 * g) =
                        make_float4(scr[tk * 33 + 4 * g], scr[tk * 33 + 4 * g + 1], scr[tk * 33 + 4 * g + 2], scr[tk * 33 + 4 * g + 3]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

extern "C" int64_t al3d_tok_patch_embed_image_bytes(void) { return 3 * 3 * 2 * 1024; }

extern "C" int al3d_tok_patch_embed_f16x3(const float* img, int B, int H, int W, const void* image, float scale, const float* bias,
                                          const float* gamma, const float* beta, float eps, float* out, void* stream)
{
    AL3D_REQUIRE(B >= 0 && H >= 1 && W >= 4 && W % 4 == 0, "al3d_tok_patch_embed_f16x3: W=%d must be a multiple of 4 (a patch row is read as aligned 16-byte pieces); other widths: al3d_tok_patch_rows_f32 + al3d_tok_linear_f16x3 + al3d_tok_layernorm_f32", W);
    if (B == 0) return AL3D_OK;
    AL3D_REQUIRE(img && image && out && (!gamma || beta), "al3d_tok_patch_embed_f16x3: null pointer");
    AL3D_REQUIRE((((uintptr_t)img | (uintptr_t)image | (uintptr_t)out) & 15) == 0, "al3d_tok_patch_embed_f16x3: img / image / out must be 16-byte aligned");
    TokPatchEmbedParams p;
    p.img = img; p.image = (const _Float16*)image; p.bias = bias; p.gamma = gamma; p.beta = beta; p.out = out;
    p.scale = scale; p.eps = eps; p.B = B; p.H = H; p.W = W; p.TH = (H + 3) / 4; p.TW = W / 4;
    p.T = (int64_t)B * p.TH * p.TW;
    AL3D_REQUIRE(p.T < ((int64_t)1 << 31), "al3d_tok_patch_embed_f16x3: too many tokens");
    p.gpw = 8;                                               // 32-token groups per wave (the weights are staged once per workgroup)
    const int64_t groups = al3d_cdiv(p.T, 32);
    const dim3 grid((unsigned)al3d_cdiv(groups, 4 * p.gpw));
    hipLaunchKernelGGL(tok_patch_embed_f16x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("tok_patch_embed_f16x3_kernel");
    return AL3D_OK;
}
