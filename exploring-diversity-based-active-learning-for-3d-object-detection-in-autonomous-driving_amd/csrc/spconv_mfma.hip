// Sparse convolution as an implicit GEMM on the fp32 matrix cores (gfx950).
//
// Same math as sp_conv_kernel (spconv.hip) -- out[o] = sum_k W[k]^T in[nbr[o][k]] with the
// fused scale/shift(+residual)(+ReLU) epilogue -- but the tile of 128 output rows is treated
// like the pixel tile of the dense conv kernel (conv2d_mfma.hip): per (kernel offset, 32-channel
// chunk) step the gathered input rows and the weight slice are register-staged into
// double-buffered, 36-float-padded LDS tiles and consumed by v_mfma_f32_32x32x2_f32.
// The gather is the A-operand address computation: row r of the tile reads feature row
// nbr[row0+r][k] (or zeros when the site has no neighbour at that offset), 128 B per row and
// chunk, so every access is a full cache line.  Offsets with no neighbour in the whole tile
// are skipped (two barriers cheaper than 16..64 MFMAs of zeros).
// Weights are pre-packed [Cout][K][Cin] so both operands are "row x contiguous-K".
#include "al3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SM_BM 128

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void sp_conv_mfma_kernel(const float* __restrict__ fin,
                                                              const int* __restrict__ nbr, int K,
                                                              const float* __restrict__ wgt,  // [COUT][K][CIN]
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ residual, int relu,
                                                              float* __restrict__ fout, int n_out)
{
    constexpr int BK = CIN < 32 ? CIN : 32;
    constexpr int LD = BK + 4;
    constexpr int QPR = BK / 4;                     // float4 pieces per row
    constexpr int RPP = 256 / QPR;                  // rows staged per pass
    constexpr int A_PASSES = SM_BM / RPP;
    constexpr int NP = COUT < 32 ? 32 : COUT;       // N padded to one MFMA tile (16-wide layers)
    constexpr int B_PASSES = (NP + RPP - 1) / RPP;
    constexpr int WN = NP >= 128 ? 64 : 32;         // columns per wave
    constexpr int WAVES_N = NP / WN;                // 2, 2, 1, 1 for COUT = 128, 64, 32, 16
    constexpr int WAVES_M = 4 / WAVES_N;            // 2, 2, 4
    constexpr int WM = SM_BM / WAVES_M;             // 64, 64, 32 rows per wave
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int KCHUNKS = CIN / BK;
    constexpr bool B_EXACT = (NP % RPP) == 0;
    __shared__ __attribute__((aligned(16))) float As[2][SM_BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][NP * LD];
    __shared__ unsigned s_mask;
    __shared__ int s_taps[32];
    __shared__ int s_ntaps;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int row0 = blockIdx.x * SM_BM;
    const int sq = tid % QPR, sr = tid / QPR;
    const int fr = lane & 31, fh = lane >> 5;

    // which kernel offsets have at least one neighbour in this tile (one coalesced pass over
    // the tile's [128][K] rulebook block)
    if (tid == 0) s_mask = 0u;
    __syncthreads();
    {
        unsigned m = 0u;
        const int rows = n_out - row0 < SM_BM ? n_out - row0 : SM_BM;
        for (int e = tid; e < SM_BM * K; e += 256) {       // tap-major rulebook: nbr[k][row]
            const int k = e / SM_BM, r = e % SM_BM;
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
    // per-wave refinement: offsets that are empty for one of this wave's 32-row sub-tiles are
    // skipped for that sub-tile (rows are in raster order, so whole sub-tiles often miss the
    // dz/dy != 0 offsets)
    unsigned wmask[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned m = 0u;
        const int row = row0 + wm * WM + i * 32 + fr;
        for (int k0 = 0; k0 < K; k0 += 2) {            // all 64 lanes run every iteration
            const int k = k0 + fh;                     // low half tests offset k0, high half k0+1
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

    float4 ra[A_PASSES], rb[B_PASSES];
    int src[A_PASSES], src_n[A_PASSES];
    int cur_tap = -1, nxt_tap = -1;
    auto fetch_idx = [&](int tap, int* dst) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int row = row0 + sr + RPP * i;
            dst[i] = row < n_out ? nbr[(int64_t)tap * n_out + row] : -1;
        }
    };
    auto load_step = [&](int step) {
        const int tap = s_taps[step / KCHUNKS], c0 = (step % KCHUNKS) * BK;
        if (tap != cur_tap) {                       // wave-uniform: new kernel offset, new gather rows
            if (tap == nxt_tap) {
#pragma unroll
                for (int i = 0; i < A_PASSES; ++i) src[i] = src_n[i];
            } else {
                fetch_idx(tap, src);
            }
            cur_tap = tap;
            // gather ids of the following offset are requested one offset early, so the
            // id -> row dependent-load chain is off the critical path
            const int ti = step / KCHUNKS + 1;
            if (ti < s_ntaps) { nxt_tap = s_taps[ti]; fetch_idx(nxt_tap, src_n); }
        }
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            ra[i] = src[i] >= 0 ? *reinterpret_cast<const float4*>(fin + (int64_t)src[i] * CIN + c0 + 4 * sq)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = sr + RPP * i;
            if (B_EXACT || n < NP)
                rb[i] = n < COUT ? *reinterpret_cast<const float4*>(wgt + ((int64_t)n * K + tap) * CIN + c0 + 4 * sq)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As[buf][(sr + RPP * i) * LD + 4 * sq]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = sr + RPP * i;
            if (B_EXACT || n < NP) *reinterpret_cast<float4*>(&Bs[buf][n * LD + 4 * sq]) = rb[i];
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
        const float* Ab = &As[buf][(wm * WM + fr) * LD + 4 * fh];
        const float* Bb = &Bs[buf][(wn * WN + fr) * LD + 4 * fh];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float av[TM][4], bv[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(Ab + i * 32 * LD + kg * 8);
                av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(Bb + j * 32 * LD + kg * 8);
                bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (!(wmask[i] >> tap_now & 1u)) continue;      // wave-uniform
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
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
                if (residual) v += residual[(int64_t)row * COUT + n];
                if (relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                fout[(int64_t)row * COUT + n] = v;
            }
        }
    }
}

#define SPM_DISPATCH(CI, CO)                                                                          \
    if (cin == CI && cout == CO) {                                                                    \
        hipLaunchKernelGGL((sp_conv_mfma_kernel<CI, CO>), dim3((unsigned)al3d_cdiv(n_out, SM_BM)),     \
                           dim3(256), 0, s, fin, nbr, K, wgt, scale, shift, residual, relu, fout, n_out); \
        AL3D_CHECK_LAUNCH("sp_conv_mfma_kernel");                                                     \
        return AL3D_OK;                                                                               \
    }

extern "C" int al3d_sp_conv_mfma_f32(const float* fin, const int* nbr, int K, const float* wgt_ock,
                                     int cin, int cout, const float* scale, const float* shift,
                                     const float* residual, int relu, float* fout, int n_out,
                                     void* stream)
{
    AL3D_REQUIRE(K >= 1 && n_out >= 0, "al3d_sp_conv_mfma_f32: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt_ock && fout, "al3d_sp_conv_mfma_f32: null pointer");
    const float* wgt = wgt_ock;
    hipStream_t s = (hipStream_t)stream;
    SPM_DISPATCH(16, 16) SPM_DISPATCH(16, 32) SPM_DISPATCH(32, 32) SPM_DISPATCH(32, 64) SPM_DISPATCH(64, 64)
    SPM_DISPATCH(64, 128) SPM_DISPATCH(128, 128)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_mfma_f32: unsupported channel pair %d -> %d", cin, cout);
}
