// Sparse convolution, wave-autonomous gather + bf16x6 MFMA (gfx950).
//
// sp_conv_bf16x6_kernel stages the gathered input rows through LDS like a dense GEMM tile; for
// the sparse layers that costs an LDS write + barrier per (offset, 16-channel) step whose MFMA
// work is tiny at 16..64 channels.  Here each wave owns 32 output rows and ALL output channels:
// its A operand never touches LDS -- every lane loads the 8 consecutive channels its MFMA
// fragment needs straight from the gathered feature row, splits them into the three bf16 pieces
// in registers and issues the six partial-product MFMAs per output tile.  Only the (pre-split)
// weights go through LDS, in slabs of several (offset, channel-group) units shared by the four
// waves of the workgroup, double-buffered, one barrier per slab.  Waves skip the offsets that
// are empty for their own 32 rows and otherwise run unsynchronised, so gather latency is hidden
// by the other waves on the SIMD instead of by a software pipeline.
#include "al3d_common.h"
#include "sp_rows.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SW_ROWS 128                 // rows per workgroup (4 waves x 32)
#define SW_PITCH 48                 // bytes per LDS weight row: 16 bf16 + 16 B pad

// Exact 3-way split of 8 floats into bf16 planes, written pair-wise so that each level is ONE
// v_cvt_pk_bf16_f32 per pair and the residuals are plain v_sub_f32 (this file is built with
// -fno-slp-vectorize: hipcc otherwise fuses the subtractions into v_pk_add_f32, which is several
// times dearer than two scalar subtractions next to MFMAs -- MI355X_MICROARCH.md, issue costs).
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sw_cvt_pk(float a, float b)
{
    const sw_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sw_bf16x2));
}
__device__ __forceinline__ void sw_split8(const float4& lo, const float4& hi, bf16x8& p0, bf16x8& p1, bf16x8& p2)
{
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = v[2 * e], x1 = v[2 * e + 1];
        h[e] = sw_cvt_pk(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, h[e] << 16);
        const float r1 = x1 - __builtin_bit_cast(float, h[e] & 0xffff0000u);
        m[e] = sw_cvt_pk(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, m[e] << 16);
        const float s1 = r1 - __builtin_bit_cast(float, m[e] & 0xffff0000u);
        l[e] = sw_cvt_pk(s0, s1);
    }
    const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), mv = make_uint4(m[0], m[1], m[2], m[3]),
                lv = make_uint4(l[0], l[1], l[2], l[3]);
    p0 = __builtin_bit_cast(bf16x8, hv);
    p1 = __builtin_bit_cast(bf16x8, mv);
    p2 = __builtin_bit_cast(bf16x8, lv);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sp_conv_wave_kernel(const float* __restrict__ fin,
                                                           const int* __restrict__ nbr, int K,
                                                           const __bf16* __restrict__ wgt,   // [3][COUT][K][CIN]
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ residual, int relu,
                                                           float* __restrict__ fout, int n_out)
{
    constexpr int KG = CIN / 16;                         // 16-channel groups per offset
    constexpr int TN = (COUT + 31) / 32;                 // 32-wide output tiles per wave
    constexpr int NROWS = COUT;                          // weight rows per unit
    constexpr int UNIT_BYTES = 3 * NROWS * SW_PITCH;
    constexpr int UPS_RAW = (14 * 1024) / UNIT_BYTES;    // units per slab (small slabs -> more workgroups per CU)
    constexpr int UPS = UPS_RAW < 1 ? 1 : (UPS_RAW > 16 ? 16 : UPS_RAW);
    constexpr int SLAB_PIECES = UPS * 3 * NROWS * 2;     // 16-byte pieces per slab
    constexpr int PASSES = (SLAB_PIECES + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char Ws[2][UPS * UNIT_BYTES + 64];
    __shared__ __attribute__((aligned(16))) unsigned char zrow[64];   // zero fragment for n >= COUT
    __shared__ unsigned s_mask;
    __shared__ int s_taps[32];
    __shared__ int s_ntaps;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int row0 = blockIdx.x * SW_ROWS;
    const int my_row = row0 + wave * 32 + fr;
    const int64_t plane = (int64_t)COUT * K * CIN;

    if (tid < 16) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;
    if (tid == 0) s_mask = 0u;
    __syncthreads();
    // offsets with a neighbour in this wave's rows (wmask) / anywhere in the workgroup (s_mask)
    unsigned wmask = 0u;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const int k = k0 + fh;
        const bool v = k < K && my_row < n_out && nbr[(int64_t)k * n_out + my_row] >= 0;
        const unsigned long long bal = __ballot(v);
        if (bal & 0xffffffffull) wmask |= 1u << k0;
        if (bal >> 32) wmask |= 1u << (k0 + 1);
    }
    wmask = __builtin_amdgcn_readfirstlane(wmask);
    if (lane == 0 && wmask) atomicOr(&s_mask, wmask);
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        const unsigned m = s_mask;
        for (int k = 0; k < K; ++k) if (m >> k & 1u) s_taps[c++] = k;
        s_ntaps = c;
    }
    __syncthreads();
    const int nunits = s_ntaps * KG;
    const int nslabs = (nunits + UPS - 1) / UPS;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- weight slab staging (all 256 threads): piece -> (unit, plane, n, half)
    uint4 rw[PASSES];
    auto load_slab = [&](int slab) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int piece = tid + 256 * q;
            const int half = piece & 1, n = (piece >> 1) % NROWS, pl = ((piece >> 1) / NROWS) % 3;
            const int uu = (piece >> 1) / (NROWS * 3);
            const int unit = slab * UPS + uu;
            if (piece < SLAB_PIECES && unit < nunits) {
                const int tap = s_taps[unit / KG], g = unit % KG;
                rw[q] = *reinterpret_cast<const uint4*>(wgt + pl * plane + ((int64_t)n * K + tap) * CIN + 16 * g + 8 * half);
            } else rw[q] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int piece = tid + 256 * q;
            if (piece >= SLAB_PIECES) continue;
            const int half = piece & 1, n = (piece >> 1) % NROWS, pl = ((piece >> 1) / NROWS) % 3;
            const int uu = (piece >> 1) / (NROWS * 3);
            *reinterpret_cast<uint4*>(&Ws[buf][uu * UNIT_BYTES + (pl * NROWS + n) * SW_PITCH + 16 * half]) = rw[q];
        }
    };

    if (nslabs > 0) { load_slab(0); store_slab(0); }
    __syncthreads();
    for (int slab = 0; slab < nslabs; ++slab) {
        const int buf = slab & 1;
        if (slab + 1 < nslabs) load_slab(slab + 1);
        // ---- this wave walks the units of the slab on its own
        const int u0 = slab * UPS, u1 = (u0 + UPS < nunits) ? u0 + UPS : nunits;
        int cur_tap = -1, src = -1;
        for (int unit = u0; unit < u1; ++unit) {
            const int tap = s_taps[unit / KG], g = unit % KG;
            if (!(wmask >> tap & 1u)) continue;                       // wave-uniform
            if (tap != cur_tap) {
                cur_tap = tap;
                src = my_row < n_out ? nbr[(int64_t)tap * n_out + my_row] : -1;
            }
            bf16x8 a0, a1, a2;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (src >= 0) {
                const float* rp = fin + (int64_t)src * CIN + 16 * g + 8 * fh;
                lo = *reinterpret_cast<const float4*>(rp);
                hi = *reinterpret_cast<const float4*>(rp + 4);
            }
            sw_split8(lo, hi, a0, a1, a2);
            const unsigned char* ub = &Ws[buf][(unit - u0) * UNIT_BYTES];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = j * 32 + fr;
                const bool live = n < COUT;
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(live ? ub + (0 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(live ? ub + (1 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(live ? ub + (2 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
            }
        }
        if (slab + 1 < nslabs) store_slab(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 32 + fr;
        if (n >= COUT) continue;
        const float sc = scale ? scale[n] : 1.0f;
        const float sh = shift ? shift[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (row >= n_out) continue;
            float v = acc[j][r] * sc + sh;
            const int64_t o = (int64_t)row * COUT + n;
            if (residual) v += residual[o];
            if (relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
            fout[o] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// v2: same decomposition, software-pipelined gather.  The v1 loop above is a dependent chain per
// unit (index load -> wait -> row load -> wait -> split -> MFMA) whose latency only other waves
// can hide.  Here every wave keeps a register ring of P units in flight: the neighbour index of
// unit u+2P and the gathered row fragment of unit u+P are requested while unit u is multiplied,
// so the counted vmcnt at each consume leaves P-1 row loads + P index loads outstanding.
__device__ __attribute__((aligned(256))) float g_sw_zero[128];   // stays zero: target of masked gathers
__device__ int g_sw_neg1 = -1;                                    // "no neighbour" for masked index loads

// ---- f16x3 arithmetic (see conv2d_f16x3.hip): x = xh + xl' * 2^-11 with xh = f16(x),
// xl' = f16((x - xh) * 2^11); x*w = xh*wh + xh*wl + xl'*(wh * 2^-11), three f16 MFMAs per tile.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sw_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sw_split8_f16(const float4& lo, const float4& hi, f16x8& ph, f16x8& pl)
{
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const sw_f32x2 x = {v[2 * e], v[2 * e + 1]};
        const sw_f16x2 xh = __builtin_convertvector(x, sw_f16x2);
        // (x - xh) * 2^11 == fma(xh, -2^11, x * 2^11) exactly (power-of-two scalings, exact residual)
        const sw_f32x2 r = {__builtin_fmaf((float)xh[0], -2048.0f, x[0] * 2048.0f),
                            __builtin_fmaf((float)xh[1], -2048.0f, x[1] * 2048.0f)};
        h[e] = __builtin_bit_cast(unsigned, xh);
        l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sw_f16x2));
    }
    ph = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
    pl = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
__device__ __forceinline__ uint4 sw_lift_down(const uint4& wh)       // 8 f16 * 2^-11 (packed multiplies)
{
    return __builtin_bit_cast(uint4, __builtin_bit_cast(f16x8, wh) * (_Float16)0.00048828125f);
}

template <int CIN, int COUT, int NW, int UPS, int P, int NPL>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 8))) void sp_conv_wave2_kernel(const float* __restrict__ fin,
                                                            const int* __restrict__ nbr, int K,
                                                            const unsigned short* __restrict__ wgt,   // [NPL][COUT][K][CIN]: 3 bf16 planes (bf16x6) or 2 f16 planes (f16x3)
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ residual, int relu,
                                                            float* __restrict__ fout, int n_out,
                                                            int64_t pitch, const unsigned* __restrict__ tmask, int io)
{
    // io: row formats (sp_rows.h; f16x3 only).  pitch: row pitch of the tap-major table (n_out for the plain rulebook, al3d_sp_table_pitch(n_out) for the tiled
    // one); tmask: per-32-row-tile tap masks of a tiled rulebook (or null: the masks are scanned from the table)
    constexpr int KG = CIN / 16;
    constexpr int TN = (COUT + 31) / 32;
    constexpr int NROWS = COUT;
    constexpr int UNIT_BYTES = 3 * NROWS * SW_PITCH;
    constexpr int NT = 64 * NW;                          // threads; the workgroup owns 32*NW output rows
    constexpr int G = P > UPS ? P / UPS : 1;             // slabs per unrolled group (ring slots stay compile-time)
    static_assert(P % UPS == 0 || UPS % P == 0, "ring depth and slab size must divide one another");
    constexpr int SLAB_PIECES = UPS * NPL * NROWS * 2;   // 16-byte pieces fetched per slab (LDS always holds 3 planes)
    constexpr int PASSES = (SLAB_PIECES + NT - 1) / NT;
    // weight slabs (main loop) and the epilogue's transposition scratch (one 32 x 32 f32 tile per wave,
    // 40-float pitch) share the same LDS: the loop's last barrier separates the two uses
    constexpr int EP_PITCH = 40;                         // floats: 4 * 40 % 64 == 32 -> the two row halves of a
                                                         // C fragment hit disjoint banks
    constexpr int WS_BYTES = 2 * (UPS * UNIT_BYTES + 64), EP_WAVE_BYTES = 32 * EP_PITCH * 4;
    // all waves transpose at once if that fits the slab storage (or 48 KB); otherwise in two rounds
    constexpr int EP_WAVES = NW * EP_WAVE_BYTES <= (WS_BYTES > 49152 ? WS_BYTES : 49152) ? NW : NW / 2;
    constexpr int EP_BYTES = EP_WAVES * EP_WAVE_BYTES;
    static_assert(EP_WAVES * EP_WAVE_BYTES <= 81920, "epilogue scratch too large");
    __shared__ __attribute__((aligned(16))) unsigned char smem[WS_BYTES > EP_BYTES ? WS_BYTES : EP_BYTES];
    unsigned char (*Ws)[UPS * UNIT_BYTES + 64] = reinterpret_cast<unsigned char (*)[UPS * UNIT_BYTES + 64]>(smem);
    __shared__ __attribute__((aligned(16))) unsigned char zrow[64];
    __shared__ unsigned s_mask;
    __shared__ int s_taps[32];
    __shared__ int s_ntaps;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware placement: blocks with equal blockIdx % 8 share an L2, so give each of the eight
    // groups one contiguous range of row tiles (rows are in raster order, neighbours of a tile
    // live in nearby tiles) instead of every eighth tile.  Bijective for any grid size.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int row0 = tile * (32 * NW);
    const int my_row = row0 + wave * 32 + fr;
    const bool row_ok = my_row < n_out;
    const int64_t plane = (int64_t)COUT * K * CIN;

    if (tid < 16) reinterpret_cast<unsigned*>(zrow)[tid] = 0u;
    if (tid == 0) s_mask = 0u;
    __syncthreads();
    // offsets with a neighbour in this wave's rows: all 14 index loads are issued before the first
    // ballot (one memory latency per wave instead of fourteen)
    unsigned wmask = 0u;
    if (tmask) {
        // tiled rulebook: one word per 32-row tile instead of scanning 27 x 32 table entries
        const int wtile = (row0 >> 5) + wave;
        wmask = (int64_t)wtile * 32 < n_out ? tmask[wtile] : 0u;
    } else {
        int nv[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int k = 2 * i + fh;
            const int* ip = (k < K && row_ok) ? nbr + (int64_t)k * pitch + my_row : &g_sw_neg1;
            nv[i] = *ip;
        }
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const unsigned long long bal = __ballot(nv[i] >= 0);
            if (bal & 0xffffffffull) wmask |= 1u << (2 * i);
            if (bal >> 32) wmask |= 1u << (2 * i + 1);
        }
    }
    wmask = __builtin_amdgcn_readfirstlane(wmask);
    if (lane == 0 && wmask) atomicOr(&s_mask, wmask);
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        const unsigned m = s_mask;
        for (int k = 0; k < K; ++k) if (m >> k & 1u) s_taps[c++] = k;
        s_ntaps = c;
        for (; c < 32; ++c) s_taps[c] = 31;             // bit 31 of wmask is never set
    }
    __syncthreads();
    const int nunits = s_ntaps * KG;
    const int nslabs = (nunits + UPS - 1) / UPS;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // every global load in the main loop is unconditional (clamped or redirected addresses): a
    // branch around a VMEM op makes hipcc fall back to vmcnt(0) at each consume, which would
    // serialise the ring.  The staged weight pieces are named scalars, not an array (hipcc
    // otherwise promotes the array to LDS and waits for the loads at once).
    static_assert(PASSES <= 3, "slab staging holds at most three 16-byte pieces per thread");
    uint4 rw0 = make_uint4(0u, 0u, 0u, 0u), rw1 = rw0, rw2 = rw0;
    auto slab_src = [&](int slab, int q) -> const uint4* {
        int piece = tid + NT * q;
        piece = piece < SLAB_PIECES ? piece : SLAB_PIECES - 1;
        const int half = piece & 1, n = (piece >> 1) % NROWS, pl = ((piece >> 1) / NROWS) % NPL;
        const int uu = (piece >> 1) / (NROWS * NPL);
        int unit = slab * UPS + uu;
        unit = unit < nunits ? unit : nunits - 1;
        const int tap = s_taps[unit / KG], g = unit % KG;
        return reinterpret_cast<const uint4*>(wgt + pl * plane + ((int64_t)n * K + tap) * CIN + 16 * g + 8 * half);
    };
    auto load_slab = [&](int slab) {
        rw0 = *slab_src(slab, 0);
        if constexpr (PASSES > 1) rw1 = *slab_src(slab, 1);
        if constexpr (PASSES > 2) rw2 = *slab_src(slab, 2);
    };
    auto store_piece = [&](int buf, int q, const uint4& v) {
        const int piece = tid + NT * q;
        if (piece >= SLAB_PIECES) return;
        const int half = piece & 1, n = (piece >> 1) % NROWS, pl = ((piece >> 1) / NROWS) % NPL;
        const int uu = (piece >> 1) / (NROWS * NPL);
        *reinterpret_cast<uint4*>(&Ws[buf][uu * UNIT_BYTES + (pl * NROWS + n) * SW_PITCH + 16 * half]) = v;
        if (NPL == 2 && pl == 0)                        // f16x3: third LDS plane = wh * 2^-11, derived once per slab
            *reinterpret_cast<uint4*>(&Ws[buf][uu * UNIT_BYTES + (2 * NROWS + n) * SW_PITCH + 16 * half]) = sw_lift_down(v);
    };
    auto store_slab = [&](int buf) {
        store_piece(buf, 0, rw0);
        if constexpr (PASSES > 1) store_piece(buf, 1, rw1);
        if constexpr (PASSES > 2) store_piece(buf, 2, rw2);
    };

    // Unit u = (u/KG)-th set bit of the workgroup's offset mask, channel group u%KG.  Three scalar
    // cursors walk that sequence (consume at u, row gather at u+P, index load at u+2P) so the loop
    // never has to look a unit up in LDS.
    struct Cursor { unsigned rem; int g; };
    const unsigned gmask = __builtin_amdgcn_readfirstlane(s_mask);
    auto cur_tap = [&](const Cursor& c) -> int {                       // -1: past the end / not needed by this wave
        const int t = c.rem ? __builtin_ctz(c.rem) : 31;
        return (wmask >> t & 1u) ? t : -1;
    };
    auto advance = [&](Cursor& c) {
        if (++c.g == KG) { c.g = 0; c.rem &= c.rem - 1u; }
    };
    auto load_idx = [&](const Cursor& c) -> int {
        const int t = cur_tap(c);
        const int* ip = (t >= 0 && row_ok) ? nbr + (int64_t)t * pitch + my_row : &g_sw_neg1;
        return *ip;                                                    // the loaded word is used as is, P units later
    };
    float4 dlo[P], dhi[P];
    int ridx[P];
    auto load_row = [&](const Cursor& c, int src, float4& lo, float4& hi) {
        const float* rp = src >= 0 ? fin + (int64_t)src * CIN + 16 * c.g + 8 * fh : g_sw_zero + 8 * fh;
        lo = *reinterpret_cast<const float4*>(rp);
        hi = *reinterpret_cast<const float4*>(rp + 4);
    };
    Cursor cc{gmask, 0}, rc{gmask, 0}, ic{gmask, 0};                 // consume / row / index cursors
#pragma unroll
    for (int s = 0; s < P; ++s) { ridx[s] = load_idx(ic); advance(ic); }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        load_row(rc, ridx[s], dlo[s], dhi[s]);
        advance(rc);
        ridx[s] = load_idx(ic);
        advance(ic);
    }
    if (nslabs > 0) { load_slab(0); store_slab(0); }
    __syncthreads();
    for (int slab0 = 0; slab0 < nslabs; slab0 += G) {                  // tail slabs past nslabs are empty
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int slab = slab0 + gi;
            const int buf = slab & 1;
            load_slab(slab + 1 < nslabs ? slab + 1 : nslabs - 1);
#pragma unroll
            for (int i = 0; i < UPS; ++i) {
                const int slot = (gi * UPS + i) % P;                    // compile-time after unrolling
                if (cur_tap(cc) >= 0) {
                    const unsigned char* ub = &Ws[buf][i * UNIT_BYTES];
                    if constexpr (NPL == 3) {
                        bf16x8 a0, a1, a2;
                        sw_split8(dlo[slot], dhi[slot], a0, a1, a2);
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = j * 32 + fr;
                            const bool live = n < COUT;
                            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(live ? ub + (0 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(live ? ub + (1 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(live ? ub + (2 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
                        }
                    } else {
                        f16x8 ah, al;
                        if (io & SP_IO_IN_PAIR) {                              // pair rows: the fragment IS the operand pair
                            ah = __builtin_bit_cast(f16x8, dlo[slot]);
                            al = __builtin_bit_cast(f16x8, dhi[slot]);
                        } else {
                            sw_split8_f16(dlo[slot], dhi[slot], ah, al);
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = j * 32 + fr;
                            const bool live = n < COUT;
                            const f16x8 wh = *reinterpret_cast<const f16x8*>(live ? ub + (0 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            const f16x8 wl = *reinterpret_cast<const f16x8*>(live ? ub + (1 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            const f16x8 wd = *reinterpret_cast<const f16x8*>(live ? ub + (2 * NROWS + n) * SW_PITCH + 16 * fh : zrow);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd, acc[j], 0, 0, 0);     // smallest first
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc[j], 0, 0, 0);
                        }
                        // hipcc otherwise emits read -> lgkmcnt(0) -> MFMA three times per tile through ONE
                        // register quad, exposing the LDS latency 3*TN times per unit: ask for all B
                        // fragment reads of the unit first, then the MFMAs
                        // (-3 % at 128 output channels, +5 % at 16: only where a unit has several tiles)
                        if constexpr (TN >= 2) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 3 * TN, 0);     // DS reads
                            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TN, 0);     // MFMAs
                        }
                    }
                }
                advance(cc);
                load_row(rc, ridx[slot], dlo[slot], dhi[slot]);
                advance(rc);
                ridx[slot] = load_idx(ic);
                advance(ic);
            }
            store_slab(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  The C fragment holds one column per lane and 16 scattered rows per register, so
    // storing it directly costs 16 dword stores (and 16 dword residual loads) per 32-column tile,
    // each touching two 64..128-byte pieces.  Instead every wave transposes the tile through its LDS
    // scratch and then moves float4s: the wave's 32 rows x COUT block is contiguous in fout (rows are
    // consecutive), so loads and stores are full lines.  BN, residual and ReLU are applied on the
    // float4s.
    float* scr = reinterpret_cast<float*>(smem) + (wave % EP_WAVES) * 32 * EP_PITCH;
    const int wrow0 = row0 + wave * 32;
#pragma unroll
    for (int round = 0; round < NW / EP_WAVES; ++round) {
        if (round > 0) __syncthreads();                               // the other half of the waves is done with the scratch
        if (wave / EP_WAVES != round) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_PITCH + fr] = acc[j][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the tile is in LDS (wave-private)
            __builtin_amdgcn_wave_barrier();
            sp_store_tile<COUT, EP_PITCH>(scr, lane, j, wrow0, n_out, scale, shift, residual, relu, fout, io);
            __builtin_amdgcn_wave_barrier();                          // scratch is rewritten by the next tile
        }
    }
}

#define SW2_DISPATCH(CI, CO, NW, UPS, P, NPL)                                                         \
    if (cin == CI && cout == CO) {                                                                    \
        hipLaunchKernelGGL((sp_conv_wave2_kernel<CI, CO, NW, UPS, P, NPL>),                           \
                           dim3((unsigned)al3d_cdiv(n_out, 32 * NW)), dim3(64 * NW), 0, s, fin, nbr, K, \
                           (const unsigned short*)wgt, scale, shift, residual, relu, fout, n_out,     \
                           (int64_t)nbr_pitch, tile_mask, io);                                        \
        AL3D_CHECK_LAUNCH("sp_conv_wave2_kernel");                                                    \
        return AL3D_OK;                                                                               \
    }

extern "C" int al3d_sp_conv_wave2_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3,
                                         int cin, int cout, const float* scale, const float* shift,
                                         const float* residual, int relu, float* fout, int n_out,
                                         void* stream)
{
    AL3D_REQUIRE(K >= 1 && K <= 27 && n_out >= 0, "al3d_sp_conv_wave2_bf16x6: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt_bf16x3 && fout, "al3d_sp_conv_wave2_bf16x6: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const void* wgt = wgt_bf16x3;
    const int nbr_pitch = n_out, io = 0;
    const unsigned* tile_mask = nullptr;
    SW2_DISPATCH(16, 16, 4, 4, 4, 3) SW2_DISPATCH(16, 32, 8, 2, 2, 3) SW2_DISPATCH(32, 32, 8, 2, 2, 3) SW2_DISPATCH(32, 64, 8, 4, 2, 3)
    SW2_DISPATCH(64, 64, 8, 4, 2, 3) SW2_DISPATCH(64, 128, 16, 2, 2, 3) SW2_DISPATCH(128, 128, 16, 2, 2, 3)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_wave2_bf16x6: unsupported channel pair %d -> %d", cin, cout);
}

static int sp_conv_wave2_f16x3_impl(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                                    const void* wgt_f16x2, int cin, int cout, const float* scale, const float* shift,
                                    const float* residual, int relu, float* fout, int n_out, int io, void* stream);

extern "C" int al3d_sp_conv_wave2_f16x3(const float* fin, const int* nbr, int K, const void* wgt_f16x2,
                                        int cin, int cout, const float* scale, const float* shift,
                                        const float* residual, int relu, float* fout, int n_out,
                                        void* stream)
{
    return sp_conv_wave2_f16x3_impl(fin, nbr, n_out, nullptr, K, wgt_f16x2, cin, cout, scale, shift, residual, relu, fout,
                                    n_out, 0, stream);
}

// the same kernel on a tiled rulebook (al3d_sp_*_table_tiles): pitched table, per-tile tap masks read instead of scanned
extern "C" int al3d_sp_conv_wave2_f16x3_tiles(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                                              int K, const void* wgt_f16x2, int cin, int cout, const float* scale,
                                              const float* shift, const float* residual, int relu, float* fout, int n_out,
                                              void* stream)
{
    AL3D_REQUIRE(tile_mask && nbr_pitch >= n_out && nbr_pitch % 256 == 0,
                 "al3d_sp_conv_wave2_f16x3_tiles: needs a tiled rulebook (pitch = al3d_sp_table_pitch(n_out), tile masks)");
    return sp_conv_wave2_f16x3_impl(fin, nbr, nbr_pitch, tile_mask, K, wgt_f16x2, cin, cout, scale, shift, residual, relu,
                                    fout, n_out, 0, stream);
}

// ... with the row formats of sp_rows.h: io bit 0 = input pair rows, bit 1 = write pair rows, bit 2 = residual pair rows
extern "C" int al3d_sp_conv_wave2_f16x3_tiles_io(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                                                 int K, const void* wgt_f16x2, int cin, int cout, const float* scale,
                                                 const float* shift, const float* residual, int relu, float* fout, int n_out,
                                                 int io, void* stream)
{
    AL3D_REQUIRE(tile_mask && nbr_pitch >= n_out && nbr_pitch % 256 == 0,
                 "al3d_sp_conv_wave2_f16x3_tiles_io: needs a tiled rulebook (pitch = al3d_sp_table_pitch(n_out), tile masks)");
    AL3D_REQUIRE(io >= 0 && io < 8 && cin % 8 == 0 && cout % 8 == 0, "al3d_sp_conv_wave2_f16x3_tiles_io: bad io flags / channels");
    return sp_conv_wave2_f16x3_impl(fin, nbr, nbr_pitch, tile_mask, K, wgt_f16x2, cin, cout, scale, shift, residual, relu,
                                    fout, n_out, io, stream);
}

static int sp_conv_wave2_f16x3_impl(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                                    const void* wgt_f16x2, int cin, int cout, const float* scale, const float* shift,
                                    const float* residual, int relu, float* fout, int n_out, int io, void* stream)
{
    AL3D_REQUIRE(K >= 1 && K <= 27 && n_out >= 0, "al3d_sp_conv_wave2_f16x3: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt_f16x2 && fout, "al3d_sp_conv_wave2_f16x3: null pointer");
    AL3D_REQUIRE(scale, "al3d_sp_conv_wave2_f16x3: scale carries the weight exponent and is required");
    hipStream_t s = (hipStream_t)stream;
    const void* wgt = wgt_f16x2;
    // f16x3 halves the MFMA time of a unit, so the 32-channel layers take four units per slab (one barrier
    // per 12 MFMAs instead of 6: -3 %); the LDS this needs is below the epilogue scratch anyway
    // 128 output channels: two 8-wave workgroups per CU (default) instead of one of 16 waves (AL3D_SW2_NW128=16, round 1's
    // choice on a quiet machine): a 16-wave workgroup at 128 VGPRs needs a CU's whole register file at once, so it cannot
    // start on a CU where any side-stream wave sits; and a workgroup walks the union of its waves' live taps, which is smaller
    // over 256 rows than over 512.  Same box, bench: 2,237 / 2,234 -> 2,262 / 2,248 frames/s, 128 -> 128 layers -3..5 %,
    // 64 -> 128 -9 % (profiles/r04_ab_sw2_nw128.txt).  Same arithmetic in the same order: bit-identical.  (4-wave workgroups
    // for the strided 16 -> 32 / 32 -> 64 layers: the first 2,261 -> 1,549 us, the layer after it 1,818 -> 2,382 -- the side
    // stream's work only moves; bench tie, not kept.)
    static const int nw128 = getenv("AL3D_SW2_NW128") ? atoi(getenv("AL3D_SW2_NW128")) : 8;
    if (nw128 != 16) { SW2_DISPATCH(64, 128, 8, 2, 2, 2) SW2_DISPATCH(128, 128, 8, 2, 2, 2) }
    SW2_DISPATCH(16, 16, 4, 4, 4, 2) SW2_DISPATCH(16, 32, 8, 4, 2, 2) SW2_DISPATCH(32, 32, 8, 4, 2, 2) SW2_DISPATCH(32, 64, 8, 4, 2, 2)
    SW2_DISPATCH(64, 64, 8, 4, 2, 2) SW2_DISPATCH(64, 128, 16, 2, 2, 2) SW2_DISPATCH(128, 128, 16, 2, 2, 2)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_wave2_f16x3: unsupported channel pair %d -> %d", cin, cout);
}

#define SW_DISPATCH(CI, CO)                                                                           \
    if (cin == CI && cout == CO) {                                                                    \
        hipLaunchKernelGGL((sp_conv_wave_kernel<CI, CO>), dim3((unsigned)al3d_cdiv(n_out, SW_ROWS)),   \
                           dim3(256), 0, s, fin, nbr, K, (const __bf16*)wgt_bf16x3, scale, shift,     \
                           residual, relu, fout, n_out);                                   \
        AL3D_CHECK_LAUNCH("sp_conv_wave_kernel");                                                     \
        return AL3D_OK;                                                                               \
    }

extern "C" int al3d_sp_conv_wave_bf16x6(const float* fin, const int* nbr, int K, const void* wgt_bf16x3,
                                        int cin, int cout, const float* scale, const float* shift,
                                        const float* residual, int relu, float* fout, int n_out,
                                        void* stream)
{
    AL3D_REQUIRE(K >= 1 && K <= 27 && n_out >= 0, "al3d_sp_conv_wave_bf16x6: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt_bf16x3 && fout, "al3d_sp_conv_wave_bf16x6: null pointer");
    hipStream_t s = (hipStream_t)stream;
    SW_DISPATCH(16, 16) SW_DISPATCH(16, 32) SW_DISPATCH(32, 32) SW_DISPATCH(32, 64) SW_DISPATCH(64, 64)
    SW_DISPATCH(64, 128) SW_DISPATCH(128, 128)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_wave_bf16x6: unsupported channel pair %d -> %d", cin, cout);
}
