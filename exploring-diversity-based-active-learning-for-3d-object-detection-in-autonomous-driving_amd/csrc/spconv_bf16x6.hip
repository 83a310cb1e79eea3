// Sparse convolution, fp32-faithful on the bf16 matrix cores ("bf16x6", see
// conv2d_bf16x6.hip for the arithmetic): same tiling, rulebook handling, empty-offset skipping
// and fused epilogue as sp_conv_mfma_kernel (spconv_mfma.hip); the gathered fp32 rows are split
// into three bf16 planes while they are staged into LDS, the weights are pre-split once into
// [3][Cout][K][Cin] bf16, and every K-chunk of 16 channels costs six
// v_mfma_f32_32x32x16_bf16 per output tile instead of eight v_mfma_f32_32x32x2_f32 at 1/16 of
// the rate.
#include "al3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define S6_BM 128
#define S6_BK 16
#define S6_LDB 48

__device__ __forceinline__ void s6_split3(float x, __bf16& a, __bf16& b, __bf16& c)
{
    a = (__bf16)x;
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void sp_conv_bf16x6_kernel(const float* __restrict__ fin,
                                                                const int* __restrict__ nbr, int K,
                                                                const __bf16* __restrict__ wgt,  // [3][COUT][K][CIN]
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ residual, int relu,
                                                                float* __restrict__ fout, int n_out)
{
    constexpr int NP = COUT < 32 ? 32 : COUT;
    constexpr int WN = NP >= 128 ? 64 : 32;
    constexpr int WAVES_N = NP / WN;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int WM = S6_BM / WAVES_M;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int KCHUNKS = CIN / S6_BK;
    constexpr int B_PIECES = NP * 2;                 // 16-byte pieces per weight plane and step
    constexpr int B_PASSES = (B_PIECES + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char As[2][3][S6_BM * S6_LDB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][NP * S6_LDB];
    __shared__ unsigned s_mask;
    __shared__ int s_taps[32];
    __shared__ int s_ntaps;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int row0 = blockIdx.x * S6_BM;
    const int aq = tid & 3, ar = tid >> 2;           // A: 4 float4 per row (16 ch), rows ar, ar+64
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t plane = (int64_t)COUT * K * CIN;

    if (tid == 0) s_mask = 0u;
    __syncthreads();
    {
        unsigned m = 0u;
        const int rows = n_out - row0 < S6_BM ? n_out - row0 : S6_BM;
        for (int e = tid; e < S6_BM * K; e += 256) {
            const int k = e / S6_BM, r = e % S6_BM;
            if (r < rows && nbr[(int64_t)k * n_out + row0 + r] >= 0) m |= 1u << k;
        }
        for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off);
        if (lane == 0 && m) atomicOr(&s_mask, m);
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        const unsigned m = s_mask;
        for (int k = 0; k < K; ++k) if (m >> k & 1u) s_taps[c++] = k;
        s_ntaps = c;
    }
    __syncthreads();
    const int nsteps = s_ntaps * KCHUNKS;
    unsigned wmask[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned m = 0u;
        const int row = row0 + wm * WM + i * 32 + fr;
        for (int k0 = 0; k0 < K; k0 += 2) {
            const int k = k0 + fh;
            const bool v = k < K && row < n_out && nbr[(int64_t)k * n_out + row] >= 0;
            const unsigned long long bal = __ballot(v);
            if (bal & 0xffffffffull) m |= 1u << k0;
            if (bal >> 32) m |= 1u << (k0 + 1);
        }
        wmask[i] = __builtin_amdgcn_readfirstlane(m);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2];
    uint4 rb[3][B_PASSES];
    int src[2], src_n[2];
    int cur_tap = -1, nxt_tap = -1;
    auto fetch_idx = [&](int tap, int* dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + ar + 64 * i;
            dst[i] = row < n_out ? nbr[(int64_t)tap * n_out + row] : -1;
        }
    };
    auto load_step = [&](int step) {
        const int tap = s_taps[step / KCHUNKS], c0 = (step % KCHUNKS) * S6_BK;
        if (tap != cur_tap) {
            if (tap == nxt_tap) { src[0] = src_n[0]; src[1] = src_n[1]; }
            else fetch_idx(tap, src);
            cur_tap = tap;
            const int ti = step / KCHUNKS + 1;
            if (ti < s_ntaps) { nxt_tap = s_taps[ti]; fetch_idx(nxt_tap, src_n); }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            ra[i] = src[i] >= 0 ? *reinterpret_cast<const float4*>(fin + (int64_t)src[i] * CIN + c0 + 4 * aq)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < B_PASSES; ++q) {
            const int piece = tid + 256 * q;          // piece -> (row n, half)
            const int n = piece >> 1, half = piece & 1;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                rb[pl][q] = (piece < B_PIECES && n < COUT)
                                ? *reinterpret_cast<const uint4*>(wgt + pl * plane + ((int64_t)n * K + tap) * CIN + c0 + 8 * half)
                                : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (ar + 64 * i) * S6_LDB + 8 * aq;
            const float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 a, b, c; s6_split3(v[e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
            *reinterpret_cast<bf16x4*>(&As[buf][0][off]) = h;
            *reinterpret_cast<bf16x4*>(&As[buf][1][off]) = m;
            *reinterpret_cast<bf16x4*>(&As[buf][2][off]) = l;
        }
#pragma unroll
        for (int q = 0; q < B_PASSES; ++q) {
            const int piece = tid + 256 * q;
            if (piece < B_PIECES) {
                const int n = piece >> 1, half = piece & 1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<uint4*>(&Bs[buf][pl][n * S6_LDB + 16 * half]) = rb[pl][q];
            }
        }
    };

    if (nsteps > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int tap_now = s_taps[step / KCHUNKS];
        if (step + 1 < nsteps) load_step(step + 1);
        bf16x8 b[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pl][(wn * WN + j * 32 + fr) * S6_LDB + 16 * fh]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (!(wmask[i] >> tap_now & 1u)) continue;          // wave-uniform
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                a[pl] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * WM + i * 32 + fr) * S6_LDB + 16 * fh]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[i][j], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * WN + j * 32 + fr;
        if (n >= COUT) continue;
        const float sc = scale ? scale[n] : 1.0f;
        const float sh = shift ? shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row >= n_out) continue;
                float v = acc[i][j][r] * sc + sh;
                const int64_t o = (int64_t)row * COUT + n;
                if (residual) v += residual[o];
                if (relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                fout[o] = v;
            }
        }
    }
}

#define S6_DISPATCH(CI, CO)                                                                           \
    if (cin == CI && cout == CO) {                                                                    \
        hipLaunchKernelGGL((sp_conv_bf16x6_kernel<CI, CO>), dim3((unsigned)al3d_cdiv(n_out, S6_BM)),   \
                           dim3(256), 0, s, fin, nbr, K, (const __bf16*)wgt_bf16x3, scale, shift,     \
                           residual, relu, fout, n_out);                                   \
        AL3D_CHECK_LAUNCH("sp_conv_bf16x6_kernel");                                                   \
        return AL3D_OK;                                                                               \
    }

extern "C" int al3d_sp_conv_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3,
                                   int cin, int cout, const float* scale, const float* shift,
                                   const float* residual, int relu, float* fout, int n_out,
                                   void* stream)
{
    AL3D_REQUIRE(K >= 1 && K <= 27 && n_out >= 0, "al3d_sp_conv_bf16x6: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt_bf16x3 && fout, "al3d_sp_conv_bf16x6: null pointer");
    hipStream_t s = (hipStream_t)stream;
    S6_DISPATCH(16, 16) S6_DISPATCH(16, 32) S6_DISPATCH(32, 32) S6_DISPATCH(32, 64) S6_DISPATCH(64, 64)
    S6_DISPATCH(64, 128) S6_DISPATCH(128, 128)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_bf16x6: unsupported channel pair %d -> %d", cin, cout);
}
