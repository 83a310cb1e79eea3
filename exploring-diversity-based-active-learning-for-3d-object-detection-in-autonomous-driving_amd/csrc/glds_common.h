// Shared pieces of the LDS-DMA sparse-convolution kernels (spconv_glds.hip, spconv_rng.hip): native vector types
// for inline-asm operands, the f16x3 split, raw LDS reads fused with their waits, counted vmcnt waits.
#pragma once
#include "al3d_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gl_f16x2 __attribute__((ext_vector_type(2)));
typedef float gl_f32x2 __attribute__((ext_vector_type(2)));
typedef float gl_f32x4 __attribute__((ext_vector_type(4)));     // native vectors: inline-asm register operands
typedef int gl_i32x4 __attribute__((ext_vector_type(4)));
typedef int gl_i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

static __device__ __attribute__((aligned(256))) float g_glds_zero[128];     // stays zero: source of masked gathers
static __device__ __attribute__((aligned(256))) int g_glds_neg1[64] = {      // "no neighbour": index source of items past the end
    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};

template <int I> struct gl_int { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void gl_static_for(F&& f)
{
    if constexpr (I < N) {
        f(gl_int<I>{});
        gl_static_for<N, I + 1>(f);
    }
}

// ---- the f16x3 pieces, the same operations as spconv_wave.hip (bit-identical results)
__device__ __forceinline__ void gl_split8_f16(const gl_f32x4& lo, const gl_f32x4& hi, f16x8& ph, f16x8& pl)
{
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const gl_f32x2 x = {v[2 * e], v[2 * e + 1]};
        const gl_f16x2 xh = __builtin_convertvector(x, gl_f16x2);
        const gl_f32x2 r = {__builtin_fmaf((float)xh[0], -2048.0f, x[0] * 2048.0f),
                            __builtin_fmaf((float)xh[1], -2048.0f, x[1] * 2048.0f)};
        h[e] = __builtin_bit_cast(unsigned, xh);
        l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gl_f16x2));
    }
    ph = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
    pl = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
__device__ __forceinline__ f16x8 gl_lift_down(const f16x8& wh)        // wh * 2^-11 (packed multiplies)
{
    return wh * (_Float16)0.00048828125f;
}

// ---- raw instructions the compiler must not reason about.  Every LDS read of the main loop is ONE asm block
// that also contains its `s_waitcnt lgkmcnt(0)`: with the wait in a separate statement hipcc is free to copy a
// destination register between the two (it did, merging the two arms of a branch) -- before the data arrived.
__device__ __forceinline__ void gl_lds_read_idx(gl_i32x4& d, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
}
__device__ __forceinline__ void gl_lds_read_idx(gl_i32x2& d, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
}
// A fragment only (the B fragments of this (tap, chunk) are already in registers)
__device__ __forceinline__ void gl_lds_read_a(gl_f32x4& lo, gl_f32x4& hi, unsigned a0, unsigned a1)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
}
// A fragment + the B fragments (wh, wl planes) of TN 32-column tiles; OFF = byte offset of the unit in the slab,
// PL = byte distance of the two planes, 32 rows of a plane = 1 KiB
template <int TN, int OFF, int PL>
__device__ __forceinline__ void gl_lds_read_ab(gl_f32x4& lo, gl_f32x4& hi, f16x8 (&wh)[TN], f16x8 (&wl)[TN], unsigned a0,
                                               unsigned a1, unsigned b)
{
    static_assert(TN == 1 || TN == 2 || TN == 4, "tile counts of the supported channel pairs");
    if constexpr (TN == 1)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\t"
                     "ds_read_b128 %2, %6 offset:%7\n\tds_read_b128 %3, %6 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi), "=&v"(wh[0]), "=&v"(wl[0])
                     : "v"(a0), "v"(a1), "v"(b), "n"(OFF), "n"(OFF + PL) : "memory");
    else if constexpr (TN == 2)
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\t"
                     "ds_read_b128 %2, %8 offset:%9\n\tds_read_b128 %3, %8 offset:%10\n\t"
                     "ds_read_b128 %4, %8 offset:%11\n\tds_read_b128 %5, %8 offset:%12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi), "=&v"(wh[0]), "=&v"(wl[0]), "=&v"(wh[1]), "=&v"(wl[1])
                     : "v"(a0), "v"(a1), "v"(b), "n"(OFF), "n"(OFF + PL), "n"(OFF + 1024), "n"(OFF + PL + 1024) : "memory");
    else
        asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %11\n\t"
                     "ds_read_b128 %2, %12 offset:%13\n\tds_read_b128 %3, %12 offset:%14\n\t"
                     "ds_read_b128 %4, %12 offset:%15\n\tds_read_b128 %5, %12 offset:%16\n\t"
                     "ds_read_b128 %6, %12 offset:%17\n\tds_read_b128 %7, %12 offset:%18\n\t"
                     "ds_read_b128 %8, %12 offset:%19\n\tds_read_b128 %9, %12 offset:%20\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi), "=&v"(wh[0]), "=&v"(wl[0]), "=&v"(wh[1]), "=&v"(wl[1]), "=&v"(wh[2]), "=&v"(wl[2]),
                       "=&v"(wh[3]), "=&v"(wl[3])
                     : "v"(a0), "v"(a1), "v"(b), "n"(OFF), "n"(OFF + PL), "n"(OFF + 1024), "n"(OFF + PL + 1024),
                       "n"(OFF + 2048), "n"(OFF + PL + 2048), "n"(OFF + 3072), "n"(OFF + PL + 3072) : "memory");
}
template <int N> __device__ __forceinline__ void gl_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

