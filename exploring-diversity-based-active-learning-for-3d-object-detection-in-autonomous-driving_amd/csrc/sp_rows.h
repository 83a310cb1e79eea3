// Row formats of the sparse encoder's activations and the shared epilogue of the f16x3 sparse kernels.
//
// "f32" rows: [n][C] float.  "pair" rows: the SAME 4 C bytes per row, but every group of 8 channels is stored as
// the two f16 planes the f16x3 arithmetic multiplies with -- 16 bytes xh[8] = f16(x), then 16 bytes
// xl'[8] = f16((x - xh) * 2^11) (conv2d_f16x3.hip's split).  A consumer's A fragment (8 channels of a row = 32
// bytes per lane) is then its MFMA operand pair as loaded: the split -- about two thirds of the VALU work of a
// gathered (row, tap) unit, repeated for each of the ~16 taps that reference a row -- happens ONCE, in the
// producer's epilogue.  Products are unchanged (the kernels never multiplied anything but xh and xl'), so a layer
// fed pair rows gives the same bits as the same layer fed the f32 rows they were split from.  What changes: a
// stored activation is xh + xl' 2^-11 (22-23 significant bits instead of 24), which the residual add and the
// final dense scatter see; |x| >= 65504 overflows f16 as before (the sweep checks finiteness).
//
// io flags of the *_io entry points: bit 0 = input rows are pair rows, bit 1 = write pair rows, bit 2 = the
// residual is pair rows.
#pragma once
#include "al3d_common.h"

#define SP_IO_IN_PAIR 1
#define SP_IO_OUT_PAIR 2
#define SP_IO_RES_PAIR 4

typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

// 8 floats -> (xh[8], xl'[8]) as two 16-byte words
__device__ __forceinline__ void sp_split8(const float (&v)[8], uint4& hi, uint4& lo)
{
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const sp_f32x2 x = {v[2 * e], v[2 * e + 1]};
        const sp_f16x2 xh = __builtin_convertvector(x, sp_f16x2);
        // (x - xh) * 2^11 == fma(xh, -2^11, x * 2^11) exactly (power-of-two scalings, exact residual)
        const sp_f32x2 r = {__builtin_fmaf((float)xh[0], -2048.0f, x[0] * 2048.0f),
                            __builtin_fmaf((float)xh[1], -2048.0f, x[1] * 2048.0f)};
        h[e] = __builtin_bit_cast(unsigned, xh);
        l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sp_f16x2));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// (xh[8], xl'[8]) -> xh + xl' * 2^-11
__device__ __forceinline__ void sp_unsplit8(const uint4& hi, const uint4& lo, float (&v)[8])
{
    const sp_f16x8 h = __builtin_bit_cast(sp_f16x8, hi), l = __builtin_bit_cast(sp_f16x8, lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf((float)l[e], 0.00048828125f, (float)h[e]);
}

// Epilogue of one 32-row x 32-column C tile that a wave has transposed into `scr` (row pitch EP_PITCH floats):
// BN scale / shift, residual, ReLU, store -- in either row format.  j = column tile, wrow0 = first row of the wave.
template <int COUT, int EP_PITCH>
__device__ __forceinline__ void sp_store_tile(const float* scr, int lane, int j, int wrow0, int n_out,
                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                              const float* __restrict__ residual, int relu, float* __restrict__ fout, int io)
{
    constexpr int live_max = 32;
    const int live = COUT - j * 32 < live_max ? COUT - j * 32 : live_max;     // live columns of this tile
    if (!(io & (SP_IO_OUT_PAIR | SP_IO_RES_PAIR))) {
        const int q = live / 4;                                                // float4s per row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            if (idx >= 32 * q) continue;
            const int rl = idx / q, c4 = (idx - rl * q) * 4;
            const int row = wrow0 + rl;
            if (row >= n_out) continue;
            const int n = j * 32 + c4;
            float4 v = *reinterpret_cast<const float4*>(scr + rl * EP_PITCH + c4);
            const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            const int64_t o = (int64_t)row * COUT + n;
            if (residual) {
                const float4 rs = *reinterpret_cast<const float4*>(residual + o);
                v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
            }
            if (relu) {                                                       // NaN propagates, like torch.relu
                v.x = v.x <= 0.f ? 0.f : v.x; v.y = v.y <= 0.f ? 0.f : v.y;
                v.z = v.z <= 0.f ? 0.f : v.z; v.w = v.w <= 0.f ? 0.f : v.w;
            }
            *reinterpret_cast<float4*>(fout + o) = v;
        }
        return;
    }
    // pair rows on either side: one lane per (row, 8-channel group) computes the group's 8 outputs and stores its
    // 32 bytes as two 16-byte halves.  (Two lanes per group, each storing one half so that every store instruction
    // is 64 consecutive pieces, measured SLOWER: the duplicated arithmetic costs more than the half-filled stores.)
    const int q = live / 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int grp = lane + 64 * i;
        if (grp >= 32 * q) continue;
        const int rl = grp / q, c8 = (grp - rl * q) * 8;
        const int row = wrow0 + rl;
        if (row >= n_out) continue;
        const int n = j * 32 + c8;
        const float4 a = *reinterpret_cast<const float4*>(scr + rl * EP_PITCH + c8);
        const float4 b = *reinterpret_cast<const float4*>(scr + rl * EP_PITCH + c8 + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (scale) {
            const float4 s0 = *reinterpret_cast<const float4*>(scale + n), s1 = *reinterpret_cast<const float4*>(scale + n + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e];
        }
        {
            float sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (shift) {
                const float4 t0 = *reinterpret_cast<const float4*>(shift + n), t1 = *reinterpret_cast<const float4*>(shift + n + 4);
                sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] + sh[e];
        }
        const int64_t o = (int64_t)row * COUT + n;                             // same offset in floats in both formats
        if (residual) {
            float r[8];
            if (io & SP_IO_RES_PAIR) {
                sp_unsplit8(*reinterpret_cast<const uint4*>(residual + o), *reinterpret_cast<const uint4*>(residual + o + 4), r);
            } else {
                const float4 ra = *reinterpret_cast<const float4*>(residual + o);
                const float4 rb = *reinterpret_cast<const float4*>(residual + o + 4);
                r[0] = ra.x; r[1] = ra.y; r[2] = ra.z; r[3] = ra.w; r[4] = rb.x; r[5] = rb.y; r[6] = rb.z; r[7] = rb.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] <= 0.f ? 0.f : v[e];        // NaN propagates, like torch.relu
        }
        if (io & SP_IO_OUT_PAIR) {
            uint4 hi, lo;
            sp_split8(v, hi, lo);
            *reinterpret_cast<uint4*>(fout + o) = hi;
            *reinterpret_cast<uint4*>(fout + o + 4) = lo;
        } else {
            *reinterpret_cast<float4*>(fout + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(fout + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}
