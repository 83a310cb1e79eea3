// Dense 2-D convolution in fp32-class arithmetic with THREE f16 matrix-core products per MAC.
//
// f16 carries 11 significand bits, so two pieces hold 22-23 of an fp32 value's 24 (bf16 needs
// three pieces and six products for the same, conv2d_bf16x6.hip).  What f16 lacks is exponent
// range; the split is arranged to spend as little of it as possible:
//
//   activation x (any |x| < 65504):  xh = f16(x)                A0
//                                    xl = f16((x - xh) * 2^11)  A1   (residual lifted by 2^11)
//   weight w, pre-scaled once per layer by a power of two so that max|ws| <= 2^14:
//                                    wh = f16(ws)               B0
//                                    wl = f16(ws - wh)          B1
//                                    wd = wh * 2^-11            B2   (derived in registers from the B0
//                                         fragment with v_pk_mul_f16; exact: wh >= 2^-3 for every
//                                         weight above 2^-17 max|w|, f16 subnormal rounding below)
//   x*ws = A0*B0 + A0*B1 + A1*B2 + (dropped xl*wl term <= 2^-24 |x ws|)
//
// All three products go into ONE fp32 accumulator (v_mfma_f32_32x32x16_f16, products exact);
// the weight scale 2^-s is folded into the per-channel BN scale on the host (exact).
// Per-product relative error <= ~3 * 2^-24, i.e. the size of one fp32 rounding, for |x| >= 2^-14.
// Below that xh is an f16 subnormal (v_cvt_f16_f32 and the gfx950 matrix core both keep them --
// measured, tools/probe_f16x3.py) with absolute error 2^-25, of which the lifted residual recovers
// eleven more bits: absolute representation error <= 2^-36 everywhere.  Measured against fp64
// (max error as a fraction of sum|a*b|, lognormal activations of typical magnitude m):
//      m = 1 .. 1e-5: 2.6e-7 .. 3.2e-7  (fp32-input MFMA kernel: 4.2e-7 .. 5.3e-7, bf16x6: 3.1e-7 .. 4.4e-7)
//      m = 1e-6: 1.4e-6 (and short K=32 sums already at m = 1e-5) -- the envelope: tensors whose
// typical magnitude is within [1e-4, 6e4].
// An activation >= 65504 turns into inf/NaN in the output, never a plausible number -- the sweep
// checks its embeddings for finiteness and names AL3D_MATH=bf16x6 (full fp32 range) as the way out.
//
// Same role, layouts and tiling as conv2d_bf16x6.hip: NHWC f32 activations in HBM, split while
// they are staged; weights pre-split into [2][Cout][taps][Cin] f16.  Half the MFMAs, 4 instead
// of 6 fragment reads per step, 2/3 of the LDS.
//
// Kernels in this file (all bit-identical to one another on the same layer):
//   conv3x3_f16x3_frag_kernel     3x3/s1/p1, halo in LDS, weights streamed in fragment order   (default)
//   conv3x3_f16x3_halo_kernel     3x3/s1/p1, halo and weights through LDS                      (AL3D_DENSE=lds)
//   conv2d_f16x3_bstream_kernel   any geometry, activation tile in LDS, weights streamed       (>= 24 steps)
//   conv2d_f16x3_kernel           any geometry, both tiles through LDS                         (short launches, deconv)
#include "al3d_common.h"
#include "sp_rows.h"
#include <stdlib.h>
#include <type_traits>

#define F3_BM 128
#define F3_BN 128
#define F3_BK 16
#define F3_LDB 48          // bytes per LDS row: 16 f16 (32 B) + 16 B pad (conflict-free b128 reads)
#define F3_TH 8
#define F3_TW 16

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// AL3D_F3_MAP=band (default) | rr: see f3_tile_of_block
static inline int f3_map_default()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("AL3D_F3_MAP"); v = !(e && e[0] == 'r'); }
    return v;
}

struct ConvF3Params {
    const float* in;        // [B, H, W, Cin] f32
    const _Float16* wgt;    // [2][Cout][taps][Cin] f16 planes (wh, wl)
    const float* scale;     // BN scale * 2^-s
    const float* shift;
    float* out;
    int B, H, W, Cin, Cout, OH, OW, ldc, coff;
    int ksize, stride, pad, relu;
    int tiles_x, tiles_y;
    int nblocks, ntiles;    // 128-channel column blocks; pixel tiles (tiles_x * tiles_y * B)
    int64_t plane;          // elements per weight plane = Cout * taps * Cin
    float* gap;             // optional [B][gap_parts][ldc]: per-workgroup channel sums of the stored values (fused GAP)
    int gap_parts;
    int xmap = f3_map_default();   // workgroup -> (pixel tile, column block) order, see f3_tile_of_block
    int io = 0;             // row formats of in / out (sp_rows.h: bit 0 = in pair pixels, bit 1 = write pair pixels); DMA kernel only
};

// 1-D grid -> (pixel tile, column block).  Blocks with equal blockIdx % 8 share an XCD and its L2,
// so the column blocks of one pixel tile get ids that differ by 8: they run on the same XCD at
// about the same time and the second one finds the tile's activations in L2 instead of HBM (the
// 1x1 layers are bandwidth-bound: 42 flop per byte).  Grid = ceil(ntiles / 8) * 8 * nblocks.
// xmap = 1 ("band", round 5): each XCD (blockIdx % 8) walks ONE contiguous band of pixel tiles in raster order, the column
// blocks of a tile back to back -- so that the tiles resident on an XCD at any moment are spatial neighbours (the 3x3 kernels'
// 6 x 34 halos overlap: rows shared with the tile above / below are L2 hits instead of second HBM reads) and the second column
// block still finds its activations in L2.  xmap = 0 ("rr"): tiles dealt round-robin over the XCDs (rounds 1-4).
__device__ __forceinline__ bool f3_tile_of_block(const ConvF3Params& p, int& tile, int& nblk)
{
    if (p.xmap) {
        const int k = blockIdx.x & 7, t = blockIdx.x >> 3;
        const int j = t / p.nblocks;
        nblk = t - j * p.nblocks;
        const int q8 = p.ntiles >> 3, r8 = p.ntiles & 7;
        tile = k * q8 + (k < r8 ? k : r8) + j;
        return j < q8 + (k < r8 ? 1 : 0);
    }
    const int id = blockIdx.x, span = 8 * p.nblocks;
    const int grp = id / span, rem = id - grp * span;
    nblk = rem >> 3;
    tile = grp * 8 + (rem & 7);
    return tile < p.ntiles;
}
static inline unsigned f3_grid(const ConvF3Params& p)
{
    return (unsigned)(al3d_cdiv(p.ntiles, 8) * 8 * p.nblocks);
}

__device__ __forceinline__ void split_act(float x, _Float16& h, _Float16& l)
{
    h = (_Float16)x;                                  // subnormal below 2^-14 (kept: see the header)
    l = (_Float16)__builtin_fmaf((float)h, -2048.0f, x * 2048.0f);   // == (x - h) * 2^11 exactly; one v_fma_mix
}

__device__ __forceinline__ void split_act4(const float4& v, f16x4& h, f16x4& l)
{
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { _Float16 a, b; split_act(e[i], a, b); h[i] = a; l[i] = b; }
}

// wd fragment = wh fragment * 2^-11 (packed f16 multiplies; the compiler emits v_pk_mul_f16)
__device__ __forceinline__ f16x8 lift_down(const f16x8& wh)
{
    return wh * (_Float16)0.00048828125f;
}

#define F3_IC(v) std::integral_constant<int, v>{}
#define F3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

static int convf3_check(const ConvF3Params& p, const char* name)
{
    AL3D_REQUIRE(p.in && p.wgt && p.out && p.scale, "%s: null pointer (scale carries the weight exponent and is required)", name);
    AL3D_REQUIRE(p.B >= 1 && p.H >= 1 && p.W >= 1 && p.Cin >= 1 && p.Cout >= 1, "%s: bad shape", name);
    AL3D_REQUIRE(p.Cin % F3_BK == 0, "%s: Cin=%d must be a multiple of %d", name, p.Cin, F3_BK);
    AL3D_REQUIRE(p.coff >= 0 && p.coff + p.Cout <= p.ldc, "%s: channel window [%d,%d) exceeds ldc=%d",
                 name, p.coff, p.coff + p.Cout, p.ldc);
    AL3D_REQUIRE(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0,
                 "%s: in/wgt must be 16-byte aligned", name);
    return AL3D_OK;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_f16x3_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(16))) unsigned char As[2][2][F3_BM * F3_LDB];   // [buf][plane]
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][2][F3_BN * F3_LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;
    const int wtaps = MODE == 0 ? taps : 4;

    const int aq = tid & 3, ar = tid >> 2;            // piece, row (0..63), +64 on pass 1
    int py[2], px[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = ar + 64 * i;
        py[i] = ty_ * F3_TH + m / F3_TW;
        px[i] = tx_ * F3_TW + m % F3_TW;
    }
    const int bq = tid & 1, br = tid >> 1;
    const int kchunks = p.Cin / F3_BK;
    const int nsteps = taps * kchunks;

    float4 ra[2];
    uint4 rb[2];
    auto load_step = [&](int step) {
        const int tap = step / kchunks, c0 = (step - tap * kchunks) * F3_BK;
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
        for (int pl = 0; pl < 2; ++pl)
            rb[pl] = n < p.Cout ? *reinterpret_cast<const uint4*>(
                                      p.wgt + pl * p.plane + ((int64_t)n * wtaps + tap0 + tap) * p.Cin + c0 + 8 * bq)
                                : make_uint4(0u, 0u, 0u, 0u);
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f16x4 h, l;
            split_act4(ra[i], h, l);
            const int off = (ar + 64 * i) * F3_LDB + 8 * aq;
            *reinterpret_cast<f16x4*>(&As[buf][0][off]) = h;
            *reinterpret_cast<f16x4*>(&As[buf][1][off]) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            *reinterpret_cast<uint4*>(&Bs[buf][pl][br * F3_LDB + 16 * bq]) = rb[pl];
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
        f16x8 a[2][2], bb[3][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a[pl][t] = *reinterpret_cast<const f16x8*>(&As[buf][pl][(wm * 64 + t * 32 + fr) * F3_LDB + 16 * fh]);
                bb[pl][t] = *reinterpret_cast<const f16x8*>(&Bs[buf][pl][(wn * 64 + t * 32 + fr) * F3_LDB + 16 * fh]);
            }
            bb[2][t] = lift_down(bb[0][t]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = F3_MFMA(a[1][i], bb[2][j], acc[i][j]);      // xl * wh   (smallest first)
                acc[i][j] = F3_MFMA(a[0][i], bb[1][j], acc[i][j]);      // xh * wl
                acc[i][j] = F3_MFMA(a[0][i], bb[0][j], acc[i][j]);      // xh * wh
            }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
        float gsum = 0.f;                                             // this lane's column: sum of the values it stores
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                if (y >= MH || x >= MW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
                gsum += v;
            }
        }
        if (p.gap) {
            // fused global average pooling: one partial per (workgroup, wave row) and channel, fixed order
            // (rows of the lane, then the other half of the fragment), reduced by gap_parts_reduce_kernel
            gsum += __shfl_xor(gsum, 32);
            if (fh == 0) {
                const int part = (((ty_ * p.tiles_x + tx_) * (MODE == 1 ? 4 : 1) + tap0) << 1) + wm;
                p.gap[((int64_t)b * p.gap_parts + part) * p.ldc + p.coff + n] = gsum;
            }
        }
    }
}

// out[b][c] = (sum over the parts, ascending) / count
__global__ __launch_bounds__(64) void gap_parts_reduce_kernel(const float* __restrict__ gap, int parts, int C, float count,
                                                               float* __restrict__ out)
{
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
#pragma unroll 16
    for (int q = 0; q < parts; ++q) s += gap[((int64_t)b * parts + q) * C + c];
    out[(int64_t)b * C + c] = s / count;
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1
// Halo-staged kernel: the 6x34-pixel input halo of a 4x32 output tile is split and staged once
// per 16-channel chunk, the nine taps read their A fragments from it at shifted rows; only the
// weight tile changes per tap.  Output tile = 4 rows x 32 columns: lane r of an M-tile is column
// r, so the 32 lanes of a fragment read 32 consecutive halo rows (conflict-free ds_read_b128 at
// a 48-byte pitch).
//
// With three MFMAs per tile pair a step holds only 12 x 32 matrix-core cycles per wave, far less
// than an L2 round trip, so -- unlike the bf16x6 kernel this grew from -- nothing may wait inside
// a step.  Global step s = chunk * 9 + tap; everything below is software-pipelined over s:
//   * weights: global load of B(s+3) is issued at the top of step s into register set (s+1)&1;
//     the set loaded one step earlier, B(s+2), is written to LDS buffer (s+2)%3 at the end of
//     step s -- two full steps between issue and use;
//   * three weight buffers (two planes each; the third operand is derived) in LDS (32-byte rows, 16-byte halves XOR-swizzled by bit 3 of the
//     row: conflict-free b128 reads and writes without padding): B(s+1) is read by the fragment
//     prefetch of step s, B(s) was read during step s-1, B(s+2) is being written;
//   * fragments of step s+1 are read into the second register set while the MFMAs of step s run;
//   * the halo is double-buffered by chunk parity: loads for chunk c+1 are issued at taps 0 and 3
//     of chunk c (two halves, to halve the staging registers), split and stored at taps 2 and 5,
//     published by those taps' barriers, first read at tap 8;
//   * one barrier per step; 9 is odd, so two chunks are unrolled to keep every register-set
//     and buffer index a compile-time constant;
//   * every global load is unconditional (out-of-image pixels, rows beyond Cout and the steps
//     past the end read a zero word / a clamped address): a branch around a VMEM op would make
//     hipcc wait for vmcnt(0) at the next use and serialise the pipeline.
#define G3_TH 4
#define G3_TW 32
#define G3_HH (G3_TH + 2)
#define G3_HW (G3_TW + 2)
#define G3_HP (G3_HH * G3_HW)          // 204 halo pixels
#define G3_BROW 32                     // bytes per weight row in LDS (16 f16, swizzled halves)

__device__ uint4 g_f3_zero16;          // 16 zero bytes (device globals are zero-initialised)

__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_halo_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2][2][G3_HP * F3_LDB];      // [chunk parity][plane] 39.2 KB
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3][2][F3_BN * G3_BROW];     // [step % 3][plane]     24.6 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fh = lane >> 5;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int y0 = ty_ * G3_TH - 1, x0 = tx_ * G3_TW - 1;       // image coords of halo (0,0)
    const int nchunks = p.Cin / F3_BK;
    const int bq = tid & 1, br = tid >> 1;
    const float* zero = reinterpret_cast<const float*>(&g_f3_zero16);

    // ---- halo: 204 pixels x 4 float4 pieces = 816 pieces over 4 passes of 256 threads
    float4 rh[2];                                    // the halo travels in two halves (8 VGPRs, not 16)
    auto load_halo = [&](int chunk, int half) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = 2 * half + k;
            const int piece = tid + 256 * i;
            const int hp = piece >> 2, q = piece & 3;
            const int iy = y0 + hp / G3_HW, ix = x0 + hp % G3_HW;
            const bool ok = hp < G3_HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* src = p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + chunk * F3_BK + 4 * q;
            rh[k] = *reinterpret_cast<const float4*>(ok ? src : zero);
        }
    };
    auto store_halo = [&](int buf, int half) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = tid + 256 * (2 * half + k);
            const int hp = piece >> 2, q = piece & 3;
            if (hp >= G3_HP) continue;
            f16x4 h, l;
            split_act4(rh[k], h, l);
            const int off = hp * F3_LDB + 8 * q;
            *reinterpret_cast<f16x4*>(&Ah[buf][0][off]) = h;
            *reinterpret_cast<f16x4*>(&Ah[buf][1][off]) = l;
        }
    };
    // ---- weights: this thread's 16-byte piece of row n0+br, three planes
    const int nb = n0 + br;
    const bool nb_ok = nb < p.Cout;
    const _Float16* wrow = nb_ok ? p.wgt + (int64_t)nb * 9 * p.Cin + 8 * bq
                                 : reinterpret_cast<const _Float16*>(zero);
    const int64_t wplane = nb_ok ? p.plane : 0;
    const int wtap = nb_ok ? p.Cin : 0, wchunk = nb_ok ? F3_BK : 0;
    uint4 rb0[2], rb1[2];                            // two register sets (named: an [2][2] array went to scratch)
    auto load_b = [&](auto set_, int chunk, int tap) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const uint4 v = *reinterpret_cast<const uint4*>(wrow + pl * wplane + tap * wtap + chunk * wchunk);
            if constexpr (decltype(set_)::value == 0) rb0[pl] = v; else rb1[pl] = v;
        }
    };
    const int bw_off = br * G3_BROW + 16 * (bq ^ ((br >> 3) & 1));
    auto store_b = [&](auto set_, int buf) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if constexpr (decltype(set_)::value == 0) *reinterpret_cast<uint4*>(&Bs[buf][pl][bw_off]) = rb0[pl];
            else *reinterpret_cast<uint4*>(&Bs[buf][pl][bw_off]) = rb1[pl];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = ((2 * wm) * G3_HW + fr) * F3_LDB + 16 * fh;
    const int b_off = (wn * 64 + fr) * G3_BROW + 16 * (fh ^ ((fr >> 3) & 1));   // +32 rows keeps bit 3
    f16x8 fa[2][2][2], fb[2][2][2];                   // [set][plane][tile]
    // fragment reads are issued plane by plane, each right after the MFMA group that last used the
    // registers of the same plane in the current set (keeps ~1.5 fragment sets live, not 2)
    auto read_a = [&](int set, int pl, int hbuf, int tap) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            fa[set][pl][t] = *reinterpret_cast<const f16x8*>(
                &Ah[hbuf][pl][a_off + ((t + ky) * G3_HW + kx) * F3_LDB]);
    };
    auto read_b = [&](int set, int pl, int bbuf) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            fb[set][pl][t] = *reinterpret_cast<const f16x8*>(&Bs[bbuf][pl][b_off + t * 32 * G3_BROW]);
    };
    auto read_frags = [&](int set, int hbuf, int tap, int bbuf) {
        read_a(set, 1, hbuf, tap); read_b(set, 0, bbuf);
        read_b(set, 1, bbuf);
        read_a(set, 0, hbuf, tap);
    };
    auto mfma_group = [&](int set, int pa, int pb) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = F3_MFMA(fa[set][pa][i], fb[set][pb][j], acc[i][j]);
    };
    auto mfma_lifted = [&](int set) {                 // xl' * (wh * 2^-11)
        f16x8 wd[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) wd[j] = lift_down(fb[set][0][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = F3_MFMA(fa[set][1][i], wd[j], acc[i][j]);
    };

    // ---- prologue: halo(0), B(0), B(1) in LDS; B(2) in flight in register set 0
    const int last = nchunks - 1;
    load_halo(0, 0);
    load_b(F3_IC(0), 0, 0);
    store_halo(0, 0);
    load_halo(0, 1);
    store_halo(0, 1);
    store_b(F3_IC(0), 0);
    load_b(F3_IC(0), 0, 1);
    store_b(F3_IC(0), 1);
    load_b(F3_IC(0), 0, 2);
    __syncthreads();
    read_frags(0, 0, 0, 0);

    // cp (chunk parity) and tap are template-level constants so that every register-set and LDS
    // buffer index is constant from the start (a loop variable, even fully unrolled, left the
    // staging arrays in scratch memory).
    auto step = [&](auto cp_, auto tap_, int chunk, int cn) {
        constexpr int cp = decltype(cp_)::value, tap = decltype(tap_)::value;
        constexpr int q = (cp + tap) & 1;             // register-set parity of this step
        // B(s+3) -> the other register set (in flight for two steps)
        if (tap < 6) load_b(F3_IC(q ^ 1), chunk, tap + 3);
        else load_b(F3_IC(q ^ 1), cn, tap - 6);
        if (tap == 0) load_halo(cn, 0);               // next chunk's halo, first half: stored at tap 2
        if (tap == 3) load_halo(cn, 1);               // second half: stored at tap 5, first read at tap 8
        __builtin_amdgcn_sched_barrier(0);            // keep the loads at the top of the step (hipcc sinks them to the stores)
        // MFMAs of step s (smallest products first: xl*wd, xh*wl, xh*wh), interleaved with the
        // fragment reads of step s+1 (halo of the next chunk after tap 8)
        constexpr int nh = tap < 8 ? cp : cp ^ 1, nt = tap < 8 ? tap + 1 : 0, nbuf = (tap + 1) % 3;
        mfma_lifted(q);
        read_a(q ^ 1, 1, nh, nt);
        mfma_group(q, 0, 1);
        read_b(q ^ 1, 1, nbuf);
        mfma_group(q, 0, 0);
        read_a(q ^ 1, 0, nh, nt); read_b(q ^ 1, 0, nbuf);
        store_b(F3_IC(q), (tap + 2) % 3);             // B(s+2), loaded during step s-1
        if (tap == 2) store_halo(cp ^ 1, 0);
        if (tap == 5) store_halo(cp ^ 1, 1);
        __syncthreads();
    };
    auto chunk_body = [&](auto cp_, int chunk) {
        const int cn = chunk < last ? chunk + 1 : chunk;         // clamped: the last chunk re-reads itself
        step(cp_, F3_IC(0), chunk, cn); step(cp_, F3_IC(1), chunk, cn); step(cp_, F3_IC(2), chunk, cn);
        step(cp_, F3_IC(3), chunk, cn); step(cp_, F3_IC(4), chunk, cn); step(cp_, F3_IC(5), chunk, cn);
        step(cp_, F3_IC(6), chunk, cn); step(cp_, F3_IC(7), chunk, cn); step(cp_, F3_IC(8), chunk, cn);
    };
    for (int chunk0 = 0; chunk0 < nchunks; chunk0 += 2) {       // nchunks is even (checked by the launcher)
        chunk_body(F3_IC(0), chunk0);
        chunk_body(F3_IC(1), chunk0 + 1);
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = ty_ * G3_TH + 2 * wm + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = tx_ * G3_TW + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (y >= p.OH || x >= p.OW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                p.out[(((int64_t)b * p.OH + y) * p.OW + x) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1, weights from L2
// The kernel above is bound by LDS bandwidth, not by the matrix cores: per step a CU moves 64 KB
// of fragment reads + 16 KB of weight writes through LDS for 24 x 4 MFMAs (ablation on MI355X,
// B=32 128x128x128->128: full 537 us; without the weight staging 440; without any staging 398;
// without the barrier 536).  Here the weights never touch LDS: they are pre-arranged in HBM in
// FRAGMENT order -- [plane][Cout/32][chunk][tap][lane][8] f16, i.e. the 1 KiB a wave needs for one
// B operand is contiguous and lane-ordered -- and every wave loads its own B fragments straight
// into registers with one coalesced global_load_dwordx4 per fragment, two steps ahead (register
// ring of three sets; the steps of a wave walk its weight stream linearly).  What is left in LDS
// is the double-buffered activation halo, so a block needs only TWO barriers per chunk of nine
// taps (one before the next halo's first store -- its buffer was last read at tap 7 of the
// previous chunk -- and one after its last store), and the four waves drift freely in between.
// LDS traffic per step and CU: 83 KB -> 35 KB; L2 -> CU traffic doubles (both M-waves of a block
// load the same weights: 32 KB per step and CU, ~40 % of the vector L1's 64 B/clk).
// SHAPE 1 (round 5): a wave owns all FOUR image rows of the tile x ONE 32-channel tile instead of two rows x two tiles, so
// every weight fragment is loaded by exactly one wave of the workgroup (half the L2 -> CU weight traffic: 11.5 instead of
// 23 TB/s chip-wide on the 256 -> 256 layers) at twice the LDS fragment reads; same MFMA sequence per accumulator.
template <int IO = 0, int SHAPE = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_frag_kernel(ConvF3Params p)
{
    constexpr int NI = SHAPE ? 4 : 2, NJ = SHAPE ? 1 : 2;      // M tiles (image rows) x N tiles (32 channels) per wave
    constexpr bool HALO_FAR = SHAPE == 2;
    constexpr int NHB = 2;                                    // halo buffers (a third one, dropping the tap-1 barrier, measured a tie)
    static_assert(SHAPE >= 0 && SHAPE <= 2, "wave shapes 0 / 1; 2 = 1 with the far halo prefetch");
    __shared__ __attribute__((aligned(16))) unsigned char Ah[NHB][2][G3_HP * F3_LDB];    // [chunk % NHB][plane] 39.2 KB (58.8)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = SHAPE ? 0 : (wave & 1), wn = wave >> 1;
    const int row0 = SHAPE ? 0 : 2 * wm, ch0 = SHAPE ? wave * 32 : wn * 64;      // the wave's first image row / channel in the tile
    const int fr = lane & 31, fh = lane >> 5;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int y0 = ty_ * G3_TH - 1, x0 = tx_ * G3_TW - 1;       // image coords of halo (0,0)
    const int nchunks = p.Cin / F3_BK;
    const int total = 9 * nchunks;
    const float* zero = reinterpret_cast<const float*>(&g_f3_zero16);

    // the next chunk's halo travels in two halves; HALO_FAR (round 5): each half has its own registers and waits FOUR steps
    // between its request and its store (two before: a step apart is ~1.5k cycles of wall time at two waves per SIMD, less than
    // an HBM round trip under load, and vmcnt retires in order, so the wait also held the weight stream back)
    float4 rh[HALO_FAR ? 4 : 2];
    auto load_halo = [&](int chunk, int half) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = tid + 256 * (2 * half + k);
            const int hp = piece >> 2, q = piece & 3;
            const int iy = y0 + hp / G3_HW, ix = x0 + hp % G3_HW;
            const bool ok = hp < G3_HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* src = p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + chunk * F3_BK + 4 * q;
            rh[(HALO_FAR ? 2 * half : 0) + k] = *reinterpret_cast<const float4*>(ok ? src : zero);
        }
    };
    auto store_halo = [&](int buf, int half) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = tid + 256 * (2 * half + k);
            const int hp = piece >> 2, q = piece & 3;
            if (hp >= G3_HP) continue;
            f16x4 h, l;
            split_act4(rh[(HALO_FAR ? 2 * half : 0) + k], h, l);
            const int off = hp * F3_LDB + 8 * q;
            *reinterpret_cast<f16x4*>(&Ah[buf][0][off]) = h;
            *reinterpret_cast<f16x4*>(&Ah[buf][1][off]) = l;
        }
    };

    // ---- this wave's weight stream: [plane][ntile][step][lane][8]; ntile = 32 output channels
    const int NT = p.Cout >> 5;
    const int nt0 = (n0 + ch0) >> 5;
    const _Float16* bsrc[2][NJ];                      // [plane][tile]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            bsrc[pl][j] = p.wgt + ((int64_t)(pl * NT + nt0 + j) * total) * 512 + lane * 8;
    f16x8 fb0[2][NJ], fb1[2][NJ], fb2[2][NJ];         // ring of three register sets [plane][tile]
    auto load_b = [&](auto ring_, int sidx) {
        constexpr int ring = decltype(ring_)::value;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(bsrc[pl][j] + (int64_t)sidx * 512);
                if constexpr (ring == 0) fb0[pl][j] = v;
                else if constexpr (ring == 1) fb1[pl][j] = v;
                else fb2[pl][j] = v;
            }
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = (row0 * G3_HW + fr) * F3_LDB + 16 * fh;
    f16x8 fa[2][2][NI];                               // [set][plane][tile]
    auto read_a = [&](int set, int pl, int hbuf, int tap) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int t = 0; t < NI; ++t)
            fa[set][pl][t] = *reinterpret_cast<const f16x8*>(
                &Ah[hbuf][pl][a_off + ((t + ky) * G3_HW + kx) * F3_LDB]);
    };
    auto mfma_step = [&](int set, const f16x8 (&fb)[2][NJ], auto&& between0, auto&& between1) {
        f16x8 wd[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) wd[j] = lift_down(fb[0][j]);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = F3_MFMA(fa[set][1][i], wd[j], acc[i][j]);       // xl' * wd
        between0();
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = F3_MFMA(fa[set][0][i], fb[1][j], acc[i][j]);    // xh * wl
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = F3_MFMA(fa[set][0][i], fb[0][j], acc[i][j]);    // xh * wh
        between1();
    };

    // ---- prologue: halo(0) in LDS, B(0), B(1) in flight
    const int last = nchunks - 1;
    load_halo(0, 0);
    load_b(F3_IC(0), 0);
    load_b(F3_IC(1), 1);
    store_halo(0, 0);
    load_halo(0, 1);
    store_halo(0, 1);
    __syncthreads();
    read_a(0, 1, 0, 0);
    read_a(0, 0, 0, 0);

    auto step = [&](auto cp_, auto tap_, int chunk, int cn, int hb, int hbn) {
        constexpr int cp = decltype(cp_)::value, tap = decltype(tap_)::value;
        constexpr int q = (cp + tap) & 1;             // A fragment set of this step
        constexpr int ring = (cp * 9 + tap) % 3;      // B register set of this step (18 % 3 == 0)
        const int s2 = chunk * 9 + tap + 2;           // B(s+2) -> the set used at step s-1
        load_b(F3_IC((ring + 2) % 3), s2 < total ? s2 : total - 1);
        if (tap == 0) load_halo(cn, 0);               // next chunk's halo, first half: stored at tap 2 (HALO_FAR: 4)
        if (tap == (HALO_FAR ? 1 : 3)) load_halo(cn, 1);     // second half: stored at tap 5, first read at tap 8
        __builtin_amdgcn_sched_barrier(0);            // keep the loads at the top of the step
        constexpr int nt = tap < 8 ? tap + 1 : 0;
        const int nh = tap < 8 ? hb : hbn;
        auto b0 = [&]() { read_a(q ^ 1, 1, nh, nt); };
        auto b1 = [&]() { read_a(q ^ 1, 0, nh, nt); };
        if constexpr (ring == 0) mfma_step(q, fb0, b0, b1);
        else if constexpr (ring == 1) mfma_step(q, fb1, b0, b1);
        else mfma_step(q, fb2, b0, b1);
        if (tap == (HALO_FAR ? 4 : 2)) store_halo(hbn, 0);
        if (tap == 5) store_halo(hbn, 1);
        // tap 1: every wave is past the previous chunk's tap 7, the last reader of halo buffer cp^1,
        // before anyone overwrites it at tap 2; tap 6: both halves stored before the reads of tap 8
        // (three halo buffers: the one being filled was last read two chunks ago, and the tap-6 barrier bounds the drift
        // of the waves to less than a chunk: the tap-1 barrier is not needed)
        if ((tap == 1 && NHB == 2) || tap == 6) __syncthreads();
    };
    auto chunk_body = [&](auto cp_, int chunk) {
        const int cn = chunk < last ? chunk + 1 : chunk;         // clamped: the last chunk re-reads itself
        const int hb = chunk % NHB, hbn = (chunk + 1) % NHB;
        step(cp_, F3_IC(0), chunk, cn, hb, hbn); step(cp_, F3_IC(1), chunk, cn, hb, hbn); step(cp_, F3_IC(2), chunk, cn, hb, hbn);
        step(cp_, F3_IC(3), chunk, cn, hb, hbn); step(cp_, F3_IC(4), chunk, cn, hb, hbn); step(cp_, F3_IC(5), chunk, cn, hb, hbn);
        step(cp_, F3_IC(6), chunk, cn, hb, hbn); step(cp_, F3_IC(7), chunk, cn, hb, hbn); step(cp_, F3_IC(8), chunk, cn, hb, hbn);
    };
    for (int chunk0 = 0; chunk0 < nchunks; chunk0 += 2) {       // nchunks is even (checked by the launcher)
        chunk_body(F3_IC(0), chunk0);
        chunk_body(F3_IC(1), chunk0 + 1);
    }

    if constexpr (IO != 1) {
        // Each image row of the wave's tile (32 pixels x 64 channels) goes through a wave-private scratch in the halo
        // storage so that a lane holds 8 consecutive channels of a pixel and stores 32 contiguous bytes with two
        // 16-byte stores (-3 % against one dword per lane and C register straight from the fragment layout; IO = 1
        // keeps that form for A/B).  IO = 2: pair pixels out (sp_rows.h; the consumers are the DMA-fed generic
        // launches, which then skip their split): the lane splits its 8 values once and stores xh[8] | xl'[8].
        __syncthreads();                               // every wave is done reading the halo buffers
        constexpr int SP = 68;
        float* scr = reinterpret_cast<float*>(&Ah[0][0][0]) + wave * (32 * SP);
        static_assert(4 * 32 * SP * 4 <= (int)sizeof(Ah), "scratch must fit the halo storage");
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int y = ty_ * G3_TH + row0 + i;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + ch0 + j * 32 + fr;
                const float sc = p.scale[n];
                const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * sc + sh;
                    if (p.relu) v = v <= 0.f ? 0.f : v;
                    scr[((r & 3) + 8 * (r >> 2) + 4 * fh) * SP + j * 32 + fr] = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            constexpr int GPP = 4 * NJ;                // 8-channel groups per pixel of the wave's row
#pragma unroll
            for (int q = 0; q < 2 * NJ; ++q) {
                const int t = lane + 64 * q, pl = t / GPP, g = t % GPP;
                const int x = tx_ * G3_TW + pl;
                if (y >= p.OH || x >= p.OW) continue;
                const float4 a = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8);
                const float4 b4 = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8 + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
                float* o = p.out + (((int64_t)b * p.OH + y) * p.OW + x) * p.ldc + p.coff + n0 + ch0 + g * 8;
                if constexpr ((IO & SP_IO_OUT_PAIR) != 0) {
                    uint4 hi, lo;
                    sp_split8(v, hi, lo);
                    *reinterpret_cast<uint4*>(o) = hi;
                    *reinterpret_cast<uint4*>(o + 4) = lo;
                } else {
                    *reinterpret_cast<float4*>(o) = a;
                    *reinterpret_cast<float4*>(o + 4) = b4;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + ch0 + j * 32 + fr;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int y = ty_ * G3_TH + row0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = tx_ * G3_TW + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (y >= p.OH || x >= p.OW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                p.out[(((int64_t)b * p.OH + y) * p.OW + x) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1 on v_mfma_f32_16x16x32_f16
// The kernel above is power-limited: its matrix-core instructions toggle enough silicon that the chip drops to
// ~1.64 GHz under it.  On random f16 operands a bare v_mfma_f32_16x16x32_f16 loop sustains 1.93
// PFLOP/s against 1.63 for v_mfma_f32_32x32x16_f16 (tools/probes/mfma_shape_probe.hip) -- the same work at less power.
// This is the same data flow on the narrower shape: a step is (tap, PAIR of 16-channel chunks) = one K=32 MFMA depth,
// the wave tile stays 2 image rows x 32 pixels x 64 channels = 4 x 4 tiles of 16 x 16, the halo of both chunks of a pair
// sits in LDS (double-buffered by pair parity), weights stream from L2 in 16x16x32 fragment order
// ([plane][Cout/16][pair][tap][lane][8], al3d_pack_f16x3_frag16) one step ahead.  The k-blocks of a fragment are
// assigned so that the two lane quarters a b128 read group mixes (q = 0,1 / 2,3) read the SAME 16-byte half of their
// halo rows from the two chunk buffers (bases 256-byte aligned): (3 * pixel + const) mod 16 over 16 distinct pixels ->
// conflict-free at the 48-byte pitch.  Per (pixel, channel) the products are still al*wd, ah*wl, ah*wh into one fp32
// accumulator, but 32 input channels deep per instruction instead of 16: NOT bit-identical to the kernels above
// (same error class; tests compare at fp32 tolerance).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define F3_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define G16_HBUF 9984                  // one (chunk, plane) halo buffer: 204 pixels x 48 B, padded to a multiple of 256

__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_frag16_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(256))) unsigned char Ah[2][2][2][G16_HBUF];        // [pair parity][chunk][plane] 78 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int lr = lane & 15, lq = lane >> 4;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int y0 = ty_ * G3_TH - 1, x0 = tx_ * G3_TW - 1;       // image coords of halo (0,0)
    const int npairs = p.Cin / 32;
    const int total = 9 * npairs;
    const float* zero = reinterpret_cast<const float*>(&g_f3_zero16);

    // ---- halo of a chunk pair: 2 x 204 pixels x 4 float4 pieces = 1632 pieces, 8 passes of 256 threads in 4 quarters
    float4 rh[2];
    auto load_halo = [&](int pair, int quarter) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = tid + 256 * (2 * quarter + k);
            const int c = piece >= 816 ? 1 : 0, rem = piece - 816 * c;
            const int hp = rem >> 2, q = rem & 3;
            const int iy = y0 + hp / G3_HW, ix = x0 + hp % G3_HW;
            const bool ok = piece < 1632 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* src = p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + (2 * pair + c) * F3_BK + 4 * q;
            rh[k] = *reinterpret_cast<const float4*>(ok ? src : zero);
        }
    };
    auto store_halo = [&](int buf, int quarter) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = tid + 256 * (2 * quarter + k);
            if (piece >= 1632) continue;
            const int c = piece >= 816 ? 1 : 0, rem = piece - 816 * c;
            const int hp = rem >> 2, q = rem & 3;
            f16x4 h, l;
            split_act4(rh[k], h, l);
            const int off = hp * F3_LDB + 8 * q;
            *reinterpret_cast<f16x4*>(&Ah[buf][c][0][off]) = h;
            *reinterpret_cast<f16x4*>(&Ah[buf][c][1][off]) = l;
        }
    };

    // ---- this wave's weight stream: [plane][ntile16][step][lane][8]
    const int NT = p.Cout >> 4;
    const int nt0 = (n0 >> 4) + wn * 4;
    const _Float16* bsrc[2];                          // [plane], tile stride = total * 512 elements
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) bsrc[pl] = p.wgt + ((int64_t)(pl * NT + nt0) * total) * 512 + lane * 8;
    const int64_t bt = (int64_t)total * 512;
    // ONE register set [plane][ntile], rolling: the fragments of 16-channel tile j for step s+1 are requested right
    // after tile j's MFMAs of step s were issued -- a full step (48 MFMAs) between request and use, half the registers
    // of a double set
    f16x8 fb[2][4];
    auto load_b1 = [&](int j, int sidx) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            fb[pl][j] = *reinterpret_cast<const f16x8*>(bsrc[pl] + j * bt + (int64_t)sidx * 512);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // A fragment of m-tile mt (image row 2*wm + (mt >> 1), pixels 16*(mt & 1) + lr): k-block lq = chunk (lq & 1),
    // 16-byte half (lq >> 1) of the pixel's 32-byte f16 row
    const int a_off = (lq & 1) * (2 * G16_HBUF) + ((2 * wm) * G3_HW + lr) * F3_LDB + 16 * (lq >> 1);
    f16x8 fa[2][2][4];                                // [set][plane][mtile]
    auto read_a = [&](int set, int pl, int hbuf, int tap) {
        const int ky = tap / 3, kx = tap % 3;
        const unsigned char* base = &Ah[hbuf][0][pl][0] + a_off;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            fa[set][pl][mt] = *reinterpret_cast<const f16x8*>(base + (((mt >> 1) + ky) * G3_HW + 16 * (mt & 1) + kx) * F3_LDB);
    };
    auto mfma_step = [&](int set, int snext, auto&& between0, auto&& between1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 wd = lift_down(fb[0][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = F3_MFMA16(fa[set][1][i], wd, acc[i][j]);          // xl' * wd
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = F3_MFMA16(fa[set][0][i], fb[1][j], acc[i][j]);    // xh * wl
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = F3_MFMA16(fa[set][0][i], fb[0][j], acc[i][j]);    // xh * wh
            load_b1(j, snext);                        // tile j of the next step into the registers just consumed
            if (j == 0) between0();
            if (j == 1) between1();
        }
    };

    // ---- prologue: halo(pair 0) in LDS, B(0) in flight
    const int last = npairs - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b1(j, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) { load_halo(0, qd); store_halo(0, qd); }
    __syncthreads();
    read_a(0, 1, 0, 0);
    read_a(0, 0, 0, 0);

    auto step = [&](auto pp_, auto tap_, int pair, int pn) {
        constexpr int pp = decltype(pp_)::value, tap = decltype(tap_)::value;
        constexpr int q = (pp + tap) & 1;             // A fragment set of this step (9 taps: odd)
        const int s1 = pair * 9 + tap + 1;            // the next step's weights (clamped at the end: harmless re-read)
        if (tap == 0 || tap == 2 || tap == 4 || tap == 6) load_halo(pn, tap >> 1);      // next pair's halo, by quarters
        __builtin_amdgcn_sched_barrier(0);            // keep the loads at the top of the step
        constexpr int nh = tap < 8 ? pp : pp ^ 1, nt = tap < 8 ? tap + 1 : 0;
        auto b0 = [&]() { read_a(q ^ 1, 1, nh, nt); };
        auto b1 = [&]() { read_a(q ^ 1, 0, nh, nt); };
        mfma_step(q, s1 < total ? s1 : total - 1, b0, b1);
        if (tap == 1 || tap == 3 || tap == 5 || tap == 7) store_halo(pp ^ 1, tap >> 1);
        // tap 0: every wave is past the previous pair's tap 7, the last reader of halo buffer pp^1, before anyone
        // overwrites it at tap 1; tap 7: all quarters stored before the reads of tap 8
        if (tap == 0 || tap == 7) __syncthreads();
    };
    auto pair_body = [&](auto pp_, int pair) {
        const int pn = pair < last ? pair + 1 : pair;            // clamped: the last pair re-reads itself
        step(pp_, F3_IC(0), pair, pn); step(pp_, F3_IC(1), pair, pn); step(pp_, F3_IC(2), pair, pn);
        step(pp_, F3_IC(3), pair, pn); step(pp_, F3_IC(4), pair, pn); step(pp_, F3_IC(5), pair, pn);
        step(pp_, F3_IC(6), pair, pn); step(pp_, F3_IC(7), pair, pn); step(pp_, F3_IC(8), pair, pn);
    };
    for (int pair0 = 0; pair0 < npairs; pair0 += 2) {            // npairs is even (checked by the launcher)
        pair_body(F3_IC(0), pair0);
        pair_body(F3_IC(1), pair0 + 1);
    }

    // C fragment of a 16 x 16 tile: lane (lr, lq) holds column lr, rows 4*lq .. 4*lq + 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + lr;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = ty_ * G3_TH + 2 * wm + (i >> 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = tx_ * G3_TW + 16 * (i & 1) + 4 * lq + r;
                if (y >= p.OH || x >= p.OW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                p.out[(((int64_t)b * p.OH + y) * p.OW + x) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ any geometry, weights from L2
// The generic kernel with the weight half of its staging removed, like the 3x3 kernel above: the
// activation tile (128 pixels x 16 channels per step; no tap reuse to keep a halo for) is still
// split once per workgroup and double-buffered in LDS, but every wave streams its B fragments from
// fragment-ordered weights -- [plane][Cout/32 (zero-padded to 128s)][tap][chunk][lane][8] -- into a
// register ring two steps ahead.  Halves the LDS traffic and the staging work of the stride-2 /
// 1x1 / deconvolution / fused-head launches.  Steps are unrolled by six (LDS buffer parity x ring
// of three) so every buffer and register-set index is a constant; all global loads unconditional.
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_f16x3_bstream_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(16))) unsigned char As[2][2][F3_BM * F3_LDB];   // [buf][plane] 24 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;
    const int wtaps = MODE == 0 ? taps : 4;
    const int kchunks = p.Cin / F3_BK;
    const int total = taps * kchunks;
    const float* zero = reinterpret_cast<const float*>(&g_f3_zero16);

    // A staging: 128 rows x 16 ch f32 = 512 float4, 2 per thread (row ar and ar + 64)
    const int aq = tid & 3, ar = tid >> 2;
    int py[2], px[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = ar + 64 * i;
        py[i] = ty_ * F3_TH + m / F3_TW;
        px[i] = tx_ * F3_TW + m % F3_TW;
    }
    float4 ra[2];
    int ltap = 0, lchunk = 0;                         // cursor of the next A load
    auto load_a = [&]() {
        const int ky = MODE == 0 ? ltap / p.ksize : 0, kx = MODE == 0 ? ltap - ky * p.ksize : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int iy, ix;
            if (MODE == 0) { iy = py[i] * p.stride - p.pad + ky; ix = px[i] * p.stride - p.pad + kx; }
            else { iy = py[i]; ix = px[i]; }
            const bool ok = py[i] < MH && px[i] < MW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* src = p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + lchunk * F3_BK + 4 * aq;
            ra[i] = *reinterpret_cast<const float4*>(ok ? src : zero);
        }
        if (ltap * kchunks + lchunk + 1 < total) { if (++lchunk == kchunks) { lchunk = 0; ++ltap; } }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f16x4 h, l;
            split_act4(ra[i], h, l);
            const int off = (ar + 64 * i) * F3_LDB + 8 * aq;
            *reinterpret_cast<f16x4*>(&As[buf][0][off]) = h;
            *reinterpret_cast<f16x4*>(&As[buf][1][off]) = l;
        }
    };
    // B stream of this wave: [plane][ntile][tap][chunk][lane][8]
    const int NTl = (p.Cout + 127) / 128 * 4;
    const int nt0 = (n0 >> 5) + wn * 2;
    const _Float16* bsrc[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bsrc[pl][j] = p.wgt + (((int64_t)(pl * NTl + nt0 + j) * wtaps + tap0) * kchunks) * 512 + lane * 8;
    f16x8 fb0[2][2], fb1[2][2], fb2[2][2];
    auto load_b = [&](auto ring_, int sidx) {
        constexpr int ring = decltype(ring_)::value;
        const int sc = sidx < total ? sidx : total - 1;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(bsrc[pl][j] + (int64_t)sc * 512);
                if constexpr (ring == 0) fb0[pl][j] = v;
                else if constexpr (ring == 1) fb1[pl][j] = v;
                else fb2[pl][j] = v;
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int a_off = (wm * 64 + fr) * F3_LDB + 16 * fh;
    auto mfma_step = [&](int buf, const f16x8 (&fb)[2][2]) {
        f16x8 a[2][2], wd[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                a[pl][t] = *reinterpret_cast<const f16x8*>(&As[buf][pl][a_off + t * 32 * F3_LDB]);
#pragma unroll
        for (int j = 0; j < 2; ++j) wd[j] = lift_down(fb[0][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = F3_MFMA(a[1][i], wd[j], acc[i][j]);          // xl' * wd   (smallest first)
                acc[i][j] = F3_MFMA(a[0][i], fb[1][j], acc[i][j]);       // xh * wl
                acc[i][j] = F3_MFMA(a[0][i], fb[0][j], acc[i][j]);       // xh * wh
            }
    };
    // step s: B(s+2) requested, A(s+1) requested, MFMAs of step s, A(s+1) split + stored, barrier
    auto step = [&](auto k_, int s) {
        constexpr int k = decltype(k_)::value;        // s % 6
        load_b(F3_IC((k + 2) % 3), s + 2);
        load_a();
        __builtin_amdgcn_sched_barrier(0);
        if (s < total) {
            if constexpr (k % 3 == 0) mfma_step(k & 1, fb0);
            else if constexpr (k % 3 == 1) mfma_step(k & 1, fb1);
            else mfma_step(k & 1, fb2);
        }
        store_a((k & 1) ^ 1);
        __syncthreads();
    };

    load_a();                                         // A(0)
    load_b(F3_IC(0), 0);
    load_b(F3_IC(1), 1);
    store_a(0);
    __syncthreads();
    for (int s0 = 0; s0 < total; s0 += 6) {
        step(F3_IC(0), s0); step(F3_IC(1), s0 + 1); step(F3_IC(2), s0 + 2);
        step(F3_IC(3), s0 + 3); step(F3_IC(4), s0 + 4); step(F3_IC(5), s0 + 5);
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
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

// planes [2][Cout][taps][Cin] -> [2][ceil(Cout/128)*4][taps][Cin/16][64][8], zero rows beyond Cout
__global__ void pack_bstream_kernel(const _Float16* __restrict__ planes, int Cout, int taps, int Cin,
                                    _Float16* __restrict__ out, int64_t count)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= count) return;
    const int nchunks = Cin / 16, NTl = (Cout + 127) / 128 * 4;
    int64_t r = o;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int chunk = r % nchunks; r /= nchunks;
    const int tap = r % taps; r /= taps;
    const int nt = r % NTl; r /= NTl;
    const int pl = (int)r;
    const int n = nt * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5) + e;
    out[o] = n < Cout ? planes[(int64_t)pl * Cout * taps * Cin + ((int64_t)n * taps + tap) * Cin + k] : (_Float16)0.0f;
}

extern "C" int64_t al3d_pack_f16x3_bstream_elems(int Cout, int taps, int Cin)
{
    if (Cout < 1 || taps < 1 || Cin < 16 || Cin % 16) return -1;
    return 2 * (int64_t)((Cout + 127) / 128 * 128) * taps * Cin;
}

extern "C" int al3d_pack_f16x3_bstream(const void* planes_f16x2, int Cout, int taps, int Cin, void* out_frag,
                                       void* stream)
{
    AL3D_REQUIRE(planes_f16x2 && out_frag, "al3d_pack_f16x3_bstream: null pointer");
    const int64_t count = al3d_pack_f16x3_bstream_elems(Cout, taps, Cin);
    AL3D_REQUIRE(count > 0, "al3d_pack_f16x3_bstream: needs Cin %% 16 == 0 (got Cout %d, taps %d, Cin %d)", Cout, taps, Cin);
    hipLaunchKernelGGL(pack_bstream_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)planes_f16x2, Cout, taps, Cin, (_Float16*)out_frag, count);
    AL3D_CHECK_LAUNCH("pack_bstream_kernel");
    return AL3D_OK;
}

extern "C" int al3d_conv2d_nhwc_f16x3_bstream(const float* in, const void* wgt_frag, const float* scale,
                                              const float* shift, float* out, int B, int H, int W, int Cin,
                                              int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                              int relu, void* stream)
{
    ConvF3Params p;
    p.gap = nullptr; p.gap_parts = 0;
    p.in = in; p.wgt = (const _Float16*)wgt_frag; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ldc = ldc; p.coff = coff; p.relu = relu;
    AL3D_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0, "al3d_conv2d_nhwc_f16x3_bstream: bad geometry");
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    AL3D_REQUIRE(p.OH >= 1 && p.OW >= 1, "al3d_conv2d_nhwc_f16x3_bstream: empty output");
    p.plane = 0;
    int rc = convf3_check(p, "al3d_conv2d_nhwc_f16x3_bstream");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(p.OW, F3_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    hipLaunchKernelGGL(conv2d_f16x3_bstream_kernel<0>, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_bstream_kernel<conv>");
    return AL3D_OK;
}

extern "C" int al3d_deconv2x2_nhwc_f16x3_bstream(const float* in, const void* wgt_frag, const float* scale,
                                                 const float* shift, float* out, int B, int H, int W, int Cin,
                                                 int Cout, int ldc, int coff, int relu, void* stream)
{
    ConvF3Params p;
    p.gap = nullptr; p.gap_parts = 0;
    p.in = in; p.wgt = (const _Float16*)wgt_frag; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 2; p.stride = 2; p.pad = 0; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = 2 * H; p.OW = 2 * W;
    p.plane = 0;
    int rc = convf3_check(p, "al3d_deconv2x2_nhwc_f16x3_bstream");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(W, F3_TW);
    p.tiles_y = (int)al3d_cdiv(H, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    hipLaunchKernelGGL(conv2d_f16x3_bstream_kernel<1>, dim3(f3_grid(p), 1, 4), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_bstream_kernel<deconv>");
    return AL3D_OK;
}

// planes [2][Cout][9][Cin] (al3d_split_f16x3) -> fragment order [2][Cout/32][Cin/16][9][64][8]
__global__ void pack_frag_kernel(const _Float16* __restrict__ planes, int Cout, int Cin, _Float16* __restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_plane = (int64_t)Cout * 9 * Cin;
    if (o >= 2 * per_plane) return;
    const int nchunks = Cin / 16;
    int64_t r = o;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int tap = r % 9; r /= 9;
    const int chunk = r % nchunks; r /= nchunks;
    const int nt = r % (Cout / 32); r /= (Cout / 32);
    const int pl = (int)r;
    const int n = nt * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5) + e;
    out[o] = planes[pl * per_plane + ((int64_t)n * 9 + tap) * Cin + k];
}

extern "C" int al3d_pack_f16x3_frag(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream)
{
    AL3D_REQUIRE(planes_f16x2 && out_frag, "al3d_pack_f16x3_frag: null pointer");
    AL3D_REQUIRE(Cout >= 128 && Cout % 128 == 0 && Cin >= 32 && Cin % 32 == 0,
                 "al3d_pack_f16x3_frag: needs Cout %% 128 == 0 and Cin %% 32 == 0 (got %d, %d)", Cout, Cin);
    const int64_t count = 2 * (int64_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)planes_f16x2, Cout, Cin, (_Float16*)out_frag);
    AL3D_CHECK_LAUNCH("pack_frag_kernel");
    return AL3D_OK;
}

// [2][Cout][9][Cin] planes -> 16x16x32 fragment order [plane][Cout/16][pair][tap][lane][8]; lane (c, q) = output channel
// 16*ntile + c, k-block q = input channels 32*pair + 16*(q & 1) + 8*(q >> 1) .. + 7 (conv3x3_f16x3_frag16_kernel)
__global__ void pack_frag16_kernel(const _Float16* __restrict__ planes, int Cout, int Cin, _Float16* __restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_plane = (int64_t)Cout * 9 * Cin;
    if (o >= 2 * per_plane) return;
    const int npairs = Cin / 32;
    int64_t r = o;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int tap = r % 9; r /= 9;
    const int pair = r % npairs; r /= npairs;
    const int nt = r % (Cout / 16); r /= (Cout / 16);
    const int pl = (int)r;
    const int q = lane >> 4;
    const int n = nt * 16 + (lane & 15), k = pair * 32 + 16 * (q & 1) + 8 * (q >> 1) + e;
    out[o] = planes[pl * per_plane + ((int64_t)n * 9 + tap) * Cin + k];
}

extern "C" int al3d_pack_f16x3_frag16(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream)
{
    AL3D_REQUIRE(planes_f16x2 && out_frag, "al3d_pack_f16x3_frag16: null pointer");
    AL3D_REQUIRE(Cout >= 128 && Cout % 128 == 0 && Cin >= 64 && Cin % 64 == 0,
                 "al3d_pack_f16x3_frag16: needs Cout %% 128 == 0 and Cin %% 64 == 0 (got %d, %d)", Cout, Cin);
    const int64_t count = 2 * (int64_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(pack_frag16_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)planes_f16x2, Cout, Cin, (_Float16*)out_frag);
    AL3D_CHECK_LAUNCH("pack_frag16_kernel");
    return AL3D_OK;
}

extern "C" int al3d_conv3x3_nhwc_f16x3_frag16(const float* in, const void* wgt_frag16, const float* scale,
                                              const float* shift, float* out, int B, int H, int W, int Cin,
                                              int Cout, int ldc, int coff, int relu, void* stream)
{
    ConvF3Params p;
    p.gap = nullptr; p.gap_parts = 0;
    p.in = in; p.wgt = (const _Float16*)wgt_frag16; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 3; p.stride = 1; p.pad = 1; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = H; p.OW = W;
    p.plane = (int64_t)Cout * 9 * Cin;
    AL3D_REQUIRE(in && wgt_frag16 && out && scale, "al3d_conv3x3_nhwc_f16x3_frag16: null pointer");
    AL3D_REQUIRE(B >= 1 && H >= 1 && W >= 1, "al3d_conv3x3_nhwc_f16x3_frag16: bad shape");
    AL3D_REQUIRE(Cout % 128 == 0 && Cin % 64 == 0 && Cout >= 128 && Cin >= 64,
                 "al3d_conv3x3_nhwc_f16x3_frag16: needs Cout %% 128 == 0 and Cin %% 64 == 0 (got %d, %d)", Cout, Cin);
    AL3D_REQUIRE(coff >= 0 && coff + Cout <= ldc, "al3d_conv3x3_nhwc_f16x3_frag16: channel window exceeds ldc=%d", ldc);
    AL3D_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)wgt_frag16 & 15) == 0,
                 "al3d_conv3x3_nhwc_f16x3_frag16: in/wgt must be 16-byte aligned");
    p.tiles_x = (int)al3d_cdiv(p.OW, G3_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, G3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = Cout / F3_BN;
    hipLaunchKernelGGL(conv3x3_f16x3_frag16_kernel, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv3x3_f16x3_frag16_kernel");
    return AL3D_OK;
}

static int conv3x3_frag_impl(const float* in, const void* wgt_frag, const float* scale, const float* shift, float* out,
                             int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu, int io, void* stream);

extern "C" int al3d_conv3x3_nhwc_f16x3_frag(const float* in, const void* wgt_frag, const float* scale,
                                            const float* shift, float* out, int B, int H, int W, int Cin,
                                            int Cout, int ldc, int coff, int relu, void* stream)
{
    return conv3x3_frag_impl(in, wgt_frag, scale, shift, out, B, H, W, Cin, Cout, ldc, coff, relu, 0, stream);
}

// ... writing pair pixels (io = 2, csrc/sp_rows.h) for a consumer on the LDS-DMA kernel; io = 0: as above
extern "C" int al3d_conv3x3_nhwc_f16x3_frag_io(const float* in, const void* wgt_frag, const float* scale,
                                               const float* shift, float* out, int B, int H, int W, int Cin,
                                               int Cout, int ldc, int coff, int relu, int io, void* stream)
{
    AL3D_REQUIRE(io == 0 || (io == 2 && ldc % 8 == 0 && coff % 8 == 0),
                 "al3d_conv3x3_nhwc_f16x3_frag_io: io must be 0 or 2 (pair output; ldc, coff multiples of 8)");
    return conv3x3_frag_impl(in, wgt_frag, scale, shift, out, B, H, W, Cin, Cout, ldc, coff, relu, io, stream);
}

static int conv3x3_frag_impl(const float* in, const void* wgt_frag, const float* scale, const float* shift, float* out,
                             int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu, int io, void* stream)
{
    ConvF3Params p;
    p.gap = nullptr; p.gap_parts = 0;
    p.in = in; p.wgt = (const _Float16*)wgt_frag; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 3; p.stride = 1; p.pad = 1; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = H; p.OW = W;
    p.plane = (int64_t)Cout * 9 * Cin;
    AL3D_REQUIRE(in && wgt_frag && out && scale, "al3d_conv3x3_nhwc_f16x3_frag: null pointer");
    AL3D_REQUIRE(B >= 1 && H >= 1 && W >= 1, "al3d_conv3x3_nhwc_f16x3_frag: bad shape");
    AL3D_REQUIRE(Cout % 128 == 0 && Cin % 32 == 0 && Cout >= 128 && Cin >= 32,
                 "al3d_conv3x3_nhwc_f16x3_frag: needs Cout %% 128 == 0 and Cin %% 32 == 0 (got %d, %d)", Cout, Cin);
    AL3D_REQUIRE(coff >= 0 && coff + Cout <= ldc, "al3d_conv3x3_nhwc_f16x3_frag: channel window exceeds ldc=%d", ldc);
    AL3D_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)wgt_frag & 15) == 0,
                 "al3d_conv3x3_nhwc_f16x3_frag: in/wgt must be 16-byte aligned");
    p.tiles_x = (int)al3d_cdiv(p.OW, G3_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, G3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = Cout / F3_BN;
    static int direct = -1;                            // AL3D_FRAG_EPI=direct: the untransposed f32 epilogue, for A/B
    if (direct < 0) { const char* e = getenv("AL3D_FRAG_EPI"); direct = e && e[0] == 'd'; }
    const bool vec_ok = ldc % 4 == 0 && coff % 4 == 0 && ((uintptr_t)out & 15) == 0;
    // wave shape: 2 (default, round 5) = four image rows x ONE 32-channel tile per wave, the next chunk's halo halves in their
    // own registers (requested at taps 0 / 1, stored at taps 4 / 5); 1 = the same wave shape with the halves sharing two registers
    // (requested at 0 / 3, stored at 2 / 5); 0 = round 1's two rows x two tiles (AL3D_FRAG_SHAPE=0|1 for A/B).  Same bits; the
    // nine `<0>` launches of the neck: 1,670 (shape 0) -> 1,607 (1) -> 1,597 us (2) on one box
    static int shape = -1;
    if (shape < 0) { const char* e = getenv("AL3D_FRAG_SHAPE"); shape = e ? atoi(e) : 2; }
    if (shape == 2) {
        if (io & SP_IO_OUT_PAIR) hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<2, 2>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
        else if (direct || !vec_ok) hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<1, 2>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<0, 2>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    }
    else if (shape) {
        if (io & SP_IO_OUT_PAIR) hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<2, 1>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
        else if (direct || !vec_ok) hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<1, 1>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv3x3_f16x3_frag_kernel<0, 1>), dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    }
    else if (io & SP_IO_OUT_PAIR) hipLaunchKernelGGL(conv3x3_f16x3_frag_kernel<2>, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    else if (direct || !vec_ok) hipLaunchKernelGGL(conv3x3_f16x3_frag_kernel<1>, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_f16x3_frag_kernel<0>, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv3x3_f16x3_frag_kernel");
    return AL3D_OK;
}

// one-off weight split: f32 [count] * 2^sexp -> f16 [2][count] (wh, wl)
__global__ void split_weights_f16_kernel(const float* __restrict__ w, int64_t count, float mul,
                                         _Float16* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float ws = w[i] * mul;                      // power of two: exact
    const _Float16 wh = (_Float16)ws;
    out[i] = wh;
    out[count + i] = (_Float16)(ws - (float)wh);
}

extern "C" int al3d_split_f16x3(const float* w, int64_t count, int scale_exp, void* out_f16x3, void* stream)
{
    AL3D_REQUIRE(w && out_f16x3 && count >= 0, "al3d_split_f16x3: bad arguments");
    AL3D_REQUIRE(scale_exp >= -100 && scale_exp <= 100, "al3d_split_f16x3: scale exponent %d out of range", scale_exp);
    if (count == 0) return AL3D_OK;
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, count, ldexpf(1.0f, scale_exp), (_Float16*)out_f16x3);
    AL3D_CHECK_LAUNCH("split_weights_f16_kernel");
    return AL3D_OK;
}


// workgroup partials per image of the fused GAP: pixel tiles of the generic kernel x 2 wave rows (x 4 deconv taps)
extern "C" int al3d_gap_parts_count(int OH, int OW, int deconv)
{
    if (deconv) return (int)(al3d_cdiv(OW / 2, F3_TW) * al3d_cdiv(OH / 2, F3_TH)) * 8;
    return (int)(al3d_cdiv(OW, F3_TW) * al3d_cdiv(OH, F3_TH)) * 2;
}

extern "C" int al3d_gap_reduce_parts_f32(const float* gap_part, int B, int parts, int C, int64_t count, float* out,
                                         void* stream)
{
    AL3D_REQUIRE(gap_part && out && B >= 1 && parts >= 1 && C >= 1 && count >= 1, "al3d_gap_reduce_parts_f32: bad arguments");
    hipLaunchKernelGGL(gap_parts_reduce_kernel, dim3((unsigned)al3d_cdiv(C, 64), (unsigned)B), dim3(64), 0,
                       (hipStream_t)stream, gap_part, parts, C, (float)count, out);
    AL3D_CHECK_LAUNCH("gap_parts_reduce_kernel");
    return AL3D_OK;
}

static int conv2d_f16x3_impl(const float* in, const void* wgt_f16x3, const float* scale,
                             const float* shift, float* out, int B, int H, int W, int Cin,
                             int Cout, int ksize, int stride, int pad, int ldc, int coff,
                             int relu, float* gap_part, int gap_parts, void* stream)
{
    ConvF3Params p;
    p.gap = gap_part; p.gap_parts = gap_parts;
    p.in = in; p.wgt = (const _Float16*)wgt_f16x3; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ldc = ldc; p.coff = coff; p.relu = relu;
    AL3D_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0, "al3d_conv2d_nhwc_f16x3: bad geometry");
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    AL3D_REQUIRE(p.OH >= 1 && p.OW >= 1, "al3d_conv2d_nhwc_f16x3: empty output");
    p.plane = (int64_t)Cout * ksize * ksize * Cin;
    int rc = convf3_check(p, "al3d_conv2d_nhwc_f16x3");
    if (rc) return rc;
    AL3D_REQUIRE(!gap_part || gap_parts >= al3d_gap_parts_count(p.OH, p.OW, 0),
                 "al3d_conv2d_nhwc_f16x3_gap: gap_parts must be at least al3d_gap_parts_count(OH, OW, 0)");
    if (!gap_part && ksize == 3 && stride == 1 && pad == 1 && Cin % (2 * F3_BK) == 0) {     // halo-staged fast path (chunk pairs)
        p.tiles_x = (int)al3d_cdiv(p.OW, G3_TW);
        p.tiles_y = (int)al3d_cdiv(p.OH, G3_TH);
        p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
        hipLaunchKernelGGL(conv3x3_f16x3_halo_kernel, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
        AL3D_CHECK_LAUNCH("conv3x3_f16x3_halo_kernel");
        return AL3D_OK;
    }
    p.tiles_x = (int)al3d_cdiv(p.OW, F3_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    hipLaunchKernelGGL(conv2d_f16x3_kernel<0>, dim3(f3_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_kernel<conv>");
    return AL3D_OK;
}

extern "C" int al3d_conv2d_nhwc_f16x3(const float* in, const void* wgt_f16x3, const float* scale,
                                      const float* shift, float* out, int B, int H, int W, int Cin,
                                      int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                      int relu, void* stream)
{
    return conv2d_f16x3_impl(in, wgt_f16x3, scale, shift, out, B, H, W, Cin, Cout, ksize, stride, pad, ldc, coff, relu,
                             nullptr, 0, stream);
}

extern "C" int al3d_conv2d_nhwc_f16x3_gap(const float* in, const void* wgt_f16x3, const float* scale,
                                          const float* shift, float* out, int B, int H, int W, int Cin,
                                          int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                          int relu, float* gap_part, int gap_parts, void* stream)
{
    AL3D_REQUIRE(gap_part, "al3d_conv2d_nhwc_f16x3_gap: null gap_part");
    return conv2d_f16x3_impl(in, wgt_f16x3, scale, shift, out, B, H, W, Cin, Cout, ksize, stride, pad, ldc, coff, relu,
                             gap_part, gap_parts, stream);
}

static int deconv2x2_f16x3_impl(const float* in, const void* wgt_f16x3, const float* scale,
                                const float* shift, float* out, int B, int H, int W, int Cin,
                                int Cout, int ldc, int coff, int relu, float* gap_part, int gap_parts, void* stream)
{
    ConvF3Params p;
    p.gap = gap_part; p.gap_parts = gap_parts;
    p.in = in; p.wgt = (const _Float16*)wgt_f16x3; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 2; p.stride = 2; p.pad = 0; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = 2 * H; p.OW = 2 * W;
    p.plane = (int64_t)Cout * 4 * Cin;
    int rc = convf3_check(p, "al3d_deconv2x2_nhwc_f16x3");
    if (rc) return rc;
    p.tiles_x = (int)al3d_cdiv(W, F3_TW);
    p.tiles_y = (int)al3d_cdiv(H, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    AL3D_REQUIRE(!gap_part || gap_parts >= al3d_gap_parts_count(p.OH, p.OW, 1),
                 "al3d_deconv2x2_nhwc_f16x3_gap: gap_parts must be at least al3d_gap_parts_count(2H, 2W, 1)");
    hipLaunchKernelGGL(conv2d_f16x3_kernel<1>, dim3(f3_grid(p), 1, 4), dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_kernel<deconv>");
    return AL3D_OK;
}

extern "C" int al3d_deconv2x2_nhwc_f16x3(const float* in, const void* wgt_f16x3, const float* scale,
                                         const float* shift, float* out, int B, int H, int W, int Cin,
                                         int Cout, int ldc, int coff, int relu, void* stream)
{
    return deconv2x2_f16x3_impl(in, wgt_f16x3, scale, shift, out, B, H, W, Cin, Cout, ldc, coff, relu, nullptr, 0, stream);
}

extern "C" int al3d_deconv2x2_nhwc_f16x3_gap(const float* in, const void* wgt_f16x3, const float* scale,
                                             const float* shift, float* out, int B, int H, int W, int Cin,
                                             int Cout, int ldc, int coff, int relu, float* gap_part, int gap_parts,
                                             void* stream)
{
    AL3D_REQUIRE(gap_part, "al3d_deconv2x2_nhwc_f16x3_gap: null gap_part");
    return deconv2x2_f16x3_impl(in, wgt_f16x3, scale, shift, out, B, H, W, Cin, Cout, ldc, coff, relu, gap_part, gap_parts,
                                stream);
}

// ------------------------------------------------------------------ any geometry, both operands by LDS-DMA
// The generic kernels above move every operand through registers on its way to LDS (global load -> split ->
// ds_write) and wait at a barrier for the slowest of 256 such round trips once per 16-channel step; the
// streamed-weight variant trades the weight half of that for four fragment loads per wave and step, which the
// texture-address unit serves at 64 B/clk -- as many cycles as the step's twelve MFMAs.  Profiled, the stride-2,
// 1x1, deconvolution and fused-head launches sat at 180-240 TFLOP/s of products against 350 for the 3x3 kernel.
//
// Here nothing passes through a register before it is an MFMA operand.  A step's two tiles -- 128 pixels x 16
// channels of RAW fp32 activations (8 KB) and 128 output channels x 16 x {wh, wl} f16 weights (8 KB, an image
// pre-packed in exactly the LDS layout) -- are fetched by `global_load_lds_dwordx4`, four DMA instructions per
// wave and step, into a ring of NS stages that runs NS-1 steps ahead; counted `s_waitcnt vmcnt` + ONE raw
// `s_barrier` per step publish a stage.  The activation split (xh, xl') happens on the fragment, in registers,
// after the ds_read -- 2 x more VALU than splitting once per workgroup, but off the critical path of the loads.
// Bank conflicts are handled on the SOURCE side of the DMA (the LDS side is lane-linear by construction): lane
// (row r, chunk s) fetches chunk s ^ f(r), the swizzle of the sparse LDS-DMA kernel (spconv_glds.hip).
// LDS reads are inline asm with their own lgkmcnt wait: hipcc cannot see that a DMA writes LDS and would
// otherwise drain vmcnt(0) before every read it cannot disambiguate.
// Same tiles, same grid, same product order as conv2d_f16x3_kernel: bit-identical results, same GAP partials.
typedef float dg_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void dg_lds_void;
typedef const __attribute__((address_space(1))) void dg_gbl_void;
__device__ __attribute__((aligned(256))) float g_dma_zero[64];     // stays zero: source of out-of-image pixels

#define DG_STAGE 16384      // bytes per stage: A 8 KB + B 8 KB
#define DG_BOFF 8192

template <int N> __device__ __forceinline__ void dg_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int dg_swz(int r) { return (r & 1) | (((r >> 3) & 1) << 1); }

__device__ __forceinline__ void dg_split8(const dg_f32x4& lo, const dg_f32x4& hi, f16x8& ph, f16x8& pl)
{
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) { _Float16 a, b; split_act(v[e], a, b); ph[e] = a; pl[e] = b; }
}

template <int MODE, int NS>
__global__ __launch_bounds__(256, NS <= 3 ? 3 : 2) void conv2d_f16x3_dma_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * DG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;
    const int wtaps = MODE == 0 ? taps : 4;
    const int kchunks = p.Cin / F3_BK;
    const int total = taps * kchunks;
    const unsigned smem_base = (unsigned)(size_t)(dg_lds_void*)smem;

    // ---- DMA side.  Wave w fetches M-tile w (pixels 32w .. 32w+31 of the 8 x 16 tile) and the 2 KB quarter w of the
    // weight tile.  Lane (jg, sg) of piece i fetches chunk sg ^ f(r) of row r = 2 jg + i.
    const int jg = lane >> 2, sg = lane & 3;
    int py[2], px[2], cg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 2 * jg + i, m = 32 * wave + r;
        py[i] = ty_ * F3_TH + m / F3_TW;
        px[i] = tx_ * F3_TW + m % F3_TW;
        cg[i] = (sg ^ dg_swz(r)) * 4;
    }
    const char* wsrc = reinterpret_cast<const char*>(p.wgt) +
                       ((int64_t)nblk * wtaps + tap0) * kchunks * 8192 + wave * 2048 + lane * 16;
    int ltap = 0, lchunk = 0;                          // cursor of the next stage to request
    auto issue = [&](int stage) {
        const int ky = MODE == 0 ? ltap / p.ksize : 0, kx = MODE == 0 ? ltap - ky * p.ksize : 0;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + stage * DG_STAGE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int iy, ix;
            if (MODE == 0) { iy = py[i] * p.stride - p.pad + ky; ix = px[i] * p.stride - p.pad + kx; }
            else { iy = py[i]; ix = px[i]; }
            const bool ok = py[i] < MH && px[i] < MW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            // 32-bit element index (the entry checks B*H*W*Cin < 2^31): keeps the select a pair of v_cndmask, no branch
            const unsigned idx = (unsigned)(((b * p.H + iy) * p.W + ix) * p.Cin + lchunk * F3_BK + cg[i]);
            const float* src = ok ? p.in + idx : g_dma_zero + cg[i];
            __builtin_amdgcn_global_load_lds((dg_gbl_void*)src, (dg_lds_void*)(size_t)(dst + i * 1024), 16, 0, 0);
        }
        const char* ws = wsrc + ((int64_t)ltap * kchunks + lchunk) * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((dg_gbl_void*)(ws + i * 1024), (dg_lds_void*)(size_t)(dst + DG_BOFF + i * 1024),
                                             16, 0, 0);
        // steps past the end re-fetch the last one (into a free stage): the DMA count per step stays 4
        if (ltap * kchunks + lchunk + 1 < total) { if (++lchunk == kchunks) { lchunk = 0; ++ltap; } }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment side: lane (fr, fh) reads chunks 2 fh, 2 fh + 1 of pixel row fr of M-tiles 2 wm, 2 wm + 1 and
    // chunk fh of weight rows wn * 64 + {0, 32} + fr of both planes
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned offA0 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh) ^ dg_swz(fr)) * 16));
    const unsigned offA1 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh + 1) ^ dg_swz(fr)) * 16));
    const unsigned offB = (unsigned)(DG_BOFF + (wn * 64 + fr) * 32 + ((fh ^ ((fr >> 3) & 1)) * 16));

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    int stage = 0, fill = NS - 1;
    for (int s = 0; s < total; ++s) {
        dg_wait_vm<4 * (NS - 2)>();                    // this wave's share of stage s has landed ...
        __builtin_amdgcn_s_barrier();                  // ... everyone's has, and stage s-1 is free
        issue(fill);
        const unsigned sb = smem_base + stage * DG_STAGE;
        dg_f32x4 a0l, a0h, a1l, a1h;
        f16x8 wh0, wl0, wh1, wl1;
        asm volatile("ds_read_b128 %0, %8\n\t"
                     "ds_read_b128 %1, %9\n\t"
                     "ds_read_b128 %2, %8 offset:2048\n\t"
                     "ds_read_b128 %3, %9 offset:2048\n\t"
                     "ds_read_b128 %4, %10\n\t"
                     "ds_read_b128 %5, %10 offset:4096\n\t"
                     "ds_read_b128 %6, %10 offset:1024\n\t"
                     "ds_read_b128 %7, %10 offset:5120\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(wh0), "=&v"(wl0), "=&v"(wh1), "=&v"(wl1)
                     : "v"(sb + offA0), "v"(sb + offA1), "v"(sb + offB) : "memory");
        f16x8 ah[2], al[2];
        dg_split8(a0l, a0h, ah[0], al[0]);
        dg_split8(a1l, a1h, ah[1], al[1]);
        const f16x8 wd0 = lift_down(wh0), wd1 = lift_down(wh1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][0] = F3_MFMA(al[i], wd0, acc[i][0]);                  // xl' * wd   (smallest first)
            acc[i][0] = F3_MFMA(ah[i], wl0, acc[i][0]);                  // xh * wl
            acc[i][0] = F3_MFMA(ah[i], wh0, acc[i][0]);                  // xh * wh
            acc[i][1] = F3_MFMA(al[i], wd1, acc[i][1]);
            acc[i][1] = F3_MFMA(ah[i], wl1, acc[i][1]);
            acc[i][1] = F3_MFMA(ah[i], wh1, acc[i][1]);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    dg_wait_vm<0>();                                   // the tail's dummy requests must not outlive the workgroup's LDS

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
        float gsum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                if (y >= MH || x >= MW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
                gsum += v;
            }
        }
        if (p.gap) {                                                  // as conv2d_f16x3_kernel
            gsum += __shfl_xor(gsum, 32);
            if (fh == 0) {
                const int part = (((ty_ * p.tiles_x + tx_) * (MODE == 1 ? 4 : 1) + tap0) << 1) + wm;
                p.gap[((int64_t)b * p.gap_parts + part) * p.ldc + p.coff + n] = gsum;
            }
        }
    }
}

// Software-pipelined form of the kernel above (the default): the fragments of step s+1 are read while the
// MFMAs of step s run, from ONE asm block that interleaves the eight ds_reads with the twelve MFMAs and ends in
// the lgkmcnt wait -- hipcc never sees a register that is still in flight.  Two operand sets alternate (the loop
// is unrolled by two), so the split of step s+1 writes registers no queued MFMA reads.  All NS stage buffers are
// in flight: a stage's buffer is refilled as soon as every wave holds its fragments in registers.
// The four accumulator chains are interleaved inside the block (dependent MFMAs are four instructions apart);
// each accumulator still receives xl'*wd, xh*wl, xh*wh in that order per step: the same bits.
struct DgOps {
    f16x8 ah[2], al[2], wh[2], wl[2], wd[2];
};

template <int MODE, int NS, int IO = 0>
__global__ __launch_bounds__(256, 2) void conv2d_f16x3_dma2_kernel(ConvF3Params p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * DG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int tile, nblk;
    if (!f3_tile_of_block(p, tile, nblk)) return;     // padding block of the last group (uniform)
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * F3_BN;
    const int MH = MODE == 0 ? p.OH : p.H, MW = MODE == 0 ? p.OW : p.W;
    const int taps = MODE == 0 ? p.ksize * p.ksize : 1;
    const int tap0 = MODE == 0 ? 0 : blockIdx.z;
    const int wtaps = MODE == 0 ? taps : 4;
    const int kchunks = p.Cin / F3_BK;
    const int total = taps * kchunks;
    const unsigned smem_base = (unsigned)(size_t)(dg_lds_void*)smem;

    const int jg = lane >> 2, sg = lane & 3;
    int py[2], px[2], cg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 2 * jg + i, m = 32 * wave + r;
        py[i] = ty_ * F3_TH + m / F3_TW;
        px[i] = tx_ * F3_TW + m % F3_TW;
        cg[i] = (sg ^ dg_swz(r)) * 4;
    }
    const char* wsrc = reinterpret_cast<const char*>(p.wgt) +
                       ((int64_t)nblk * wtaps + tap0) * kchunks * 8192 + wave * 2048 + lane * 16;
    int ltap = 0, lchunk = 0;                          // cursor of the next stage to request
    // the pixel part of a lane's two source addresses changes only with the tap: recomputed when a tap starts
    // (wave-uniform branch), then one 64-bit multiply-add per request.  Out-of-image pixels: zero row, stride 0.
    const char* abase[2];
    unsigned ainc[2];
    auto issue = [&](int stage) {
        if (lchunk == 0) {
            const int ky = MODE == 0 ? ltap / p.ksize : 0, kx = MODE == 0 ? ltap - ky * p.ksize : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int iy, ix;
                if (MODE == 0) { iy = py[i] * p.stride - p.pad + ky; ix = px[i] * p.stride - p.pad + kx; }
                else { iy = py[i]; ix = px[i]; }
                const bool ok = py[i] < MH && px[i] < MW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                // 32-bit element index (the entry checks B*H*W*Cin < 2^31)
                const unsigned idx = (unsigned)(((b * p.H + iy) * p.W + ix) * p.Cin + cg[i]);
                abase[i] = reinterpret_cast<const char*>(ok ? p.in + idx : g_dma_zero + cg[i]);
                ainc[i] = ok ? 4u * F3_BK : 0u;
            }
        }
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + stage * DG_STAGE + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((dg_gbl_void*)(abase[i] + (size_t)lchunk * ainc[i]),
                                             (dg_lds_void*)(size_t)(dst + i * 1024), 16, 0, 0);
        const char* ws = wsrc + ((int64_t)ltap * kchunks + lchunk) * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((dg_gbl_void*)(ws + i * 1024), (dg_lds_void*)(size_t)(dst + DG_BOFF + i * 1024),
                                             16, 0, 0);
        // steps past the end re-fetch the last one (into a free stage): the DMA count per step stays 4
        if (ltap * kchunks + lchunk + 1 < total) { if (++lchunk == kchunks) { lchunk = 0; ++ltap; } }
    };

    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }

    const int fr = lane & 31, fh = lane >> 5;
    const unsigned offA0 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh) ^ dg_swz(fr)) * 16));
    const unsigned offA1 = (unsigned)(wm * 4096 + (fr & 1) * 1024 + (fr >> 1) * 64 + (((2 * fh + 1) ^ dg_swz(fr)) * 16));
    const unsigned offB = (unsigned)(DG_BOFF + (wn * 64 + fr) * 32 + ((fh ^ ((fr >> 3) & 1)) * 16));

    DgOps A, B;
    constexpr bool in_pair = (IO & SP_IO_IN_PAIR) != 0;                      // row formats are compile-time: no branch in the loop
    auto finish = [&](DgOps& o, const dg_f32x4& r0l, const dg_f32x4& r0h, const dg_f32x4& r1l, const dg_f32x4& r1h) {
        if constexpr (in_pair) {                         // pair pixels: the 32 bytes of a fragment ARE (xh[8], xl'[8])
            o.ah[0] = __builtin_bit_cast(f16x8, r0l); o.al[0] = __builtin_bit_cast(f16x8, r0h);
            o.ah[1] = __builtin_bit_cast(f16x8, r1l); o.al[1] = __builtin_bit_cast(f16x8, r1h);
        } else {
            dg_split8(r0l, r0h, o.ah[0], o.al[0]);
            dg_split8(r1l, r1h, o.ah[1], o.al[1]);
        }
        o.wd[0] = lift_down(o.wh[0]);
        o.wd[1] = lift_down(o.wh[1]);
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) issue(s);
    {
        dg_wait_vm<4 * (NS - 1)>();                    // stage 0
        __builtin_amdgcn_s_barrier();
        dg_f32x4 r0l, r0h, r1l, r1h;
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
    int nstage = 1 % NS, fill = 0;
    // one step: MFMAs on `cur`, fragments of the next stage -> `nxt`
    auto step = [&](DgOps& cur, DgOps& nxt) {
        dg_wait_vm<4 * (NS - 2)>();                    // this wave's share of the next stage has landed ...
        __builtin_amdgcn_s_barrier();                  // ... everyone's has; every wave holds the current stage in registers
        issue(fill);                                   // -> the current stage's buffer
        const unsigned sb = smem_base + nstage * DG_STAGE;
        dg_f32x4 r0l, r0h, r1l, r1h;
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
        nstage = nstage + 1 == NS ? 0 : nstage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    };
    for (int s = 0; s < total; s += 2) {
        step(A, B);
        if (s + 1 < total) step(B, A);
    }
    dg_wait_vm<0>();                                   // the tail's dummy requests must not outlive the workgroup's LDS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the VALU reads them

    const f32x16* accp[2][2] = {{&c00, &c01}, {&c10, &c11}};
    if constexpr ((IO & 4) == 0) {
        // BN / ReLU / GAP sums in the C layout, then each 32-pixel x 64-channel half of the wave's tile goes through a
        // wave-private LDS scratch (the stage ring, free after the barrier) so that a lane holds consecutive channels of
        // a pixel and stores 16-byte pieces (IO & 4: the untransposed dword stores below, for layouts that are not
        // 16-byte aligned).  Pair pixels out (IO & 2, sp_rows.h): 8 channels per lane, split once, xh[8] | xl'[8]
        __syncthreads();                               // every wave is done reading the stage ring
        constexpr int SP = 68;                         // floats per scratch row (64 + 4: conflict-free b128 reads)
        float* scr = reinterpret_cast<float*>(smem) + wave * (32 * SP);
        float gs[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + fr;
                const bool nok = n < p.Cout;
                const float sc = nok ? p.scale[n] : 0.f;
                const float sh = (nok && p.shift) ? p.shift[n] : 0.0f;
                const f32x16& acc = *accp[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int m = wm * 64 + i * 32 + ml;
                    const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                    float v = acc[r] * sc + sh;
                    if (p.relu) v = v <= 0.f ? 0.f : v;
                    scr[ml * SP + j * 32 + fr] = v;
                    if (y < MH && x < MW) gs[j] += v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if constexpr ((IO & SP_IO_OUT_PAIR) != 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = lane + 64 * q, pl = t >> 3, g = t & 7;
                    const int m = wm * 64 + i * 32 + pl;
                    const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                    const int n = n0 + wn * 64 + g * 8;
                    if (y >= MH || x >= MW || n >= p.Cout) continue;
                    const float4 a = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8);
                    const float4 b4 = *reinterpret_cast<const float4*>(scr + pl * SP + g * 8 + 4);
                    const float v[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
                    uint4 hi, lo;
                    sp_split8(v, hi, lo);
                    int oy = y, ox = x;
                    if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                    float* o = p.out + (((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n;
                    *reinterpret_cast<uint4*>(o) = hi;
                    *reinterpret_cast<uint4*>(o + 4) = lo;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {            // f32 pixels: one float4 per lane and trip (Cout % 4 == 0)
                    const int t = lane + 64 * q, pl = t >> 4, g = t & 15;
                    const int m = wm * 64 + i * 32 + pl;
                    const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                    const int n = n0 + wn * 64 + g * 4;
                    if (y >= MH || x >= MW || n >= p.Cout) continue;
                    int oy = y, ox = x;
                    if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                    *reinterpret_cast<float4*>(p.out + (((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n) =
                        *reinterpret_cast<const float4*>(scr + pl * SP + g * 4);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.gap) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + fr;
                float gsum = gs[j];
                gsum += __shfl_xor(gsum, 32);
                if (fh == 0 && n < p.Cout) {
                    const int part = (((ty_ * p.tiles_x + tx_) * (MODE == 1 ? 4 : 1) + tap0) << 1) + wm;
                    p.gap[((int64_t)b * p.gap_parts + part) * p.ldc + p.coff + n] = gsum;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n];
        const float sh = p.shift ? p.shift[n] : 0.0f;
        float gsum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x16& acc = *accp[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int y = ty_ * F3_TH + m / F3_TW, x = tx_ * F3_TW + m % F3_TW;
                if (y >= MH || x >= MW) continue;
                float v = acc[r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
                gsum += v;
            }
        }
        if (p.gap) {                                                  // as conv2d_f16x3_kernel
            gsum += __shfl_xor(gsum, 32);
            if (fh == 0) {
                const int part = (((ty_ * p.tiles_x + tx_) * (MODE == 1 ? 4 : 1) + tap0) << 1) + wm;
                p.gap[((int64_t)b * p.gap_parts + part) * p.ldc + p.coff + n] = gsum;
            }
        }
    }
}

// planes [2][Cout][taps][Cin] -> [ceil(Cout/128)][taps][Cin/16][2 planes][128 rows][2 chunks of 8 f16]; chunk c of
// row n sits at position c ^ ((n >> 3) & 1); rows >= Cout zero.  One (block, tap, chunk) = the 8 KB a stage holds.
__global__ void pack_dma_kernel(const _Float16* __restrict__ planes, int Cout, int taps, int Cin,
                                _Float16* __restrict__ out, int64_t count)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= count) return;
    const int nchunks = Cin / 16;
    int64_t r = o;
    const int e = r % 8; r /= 8;
    const int pos = r % 2; r /= 2;
    const int nl = r % 128; r /= 128;
    const int pl = r % 2; r /= 2;
    const int chunk = r % nchunks; r /= nchunks;
    const int tap = r % taps; r /= taps;
    const int n = (int)r * 128 + nl;
    const int c = pos ^ ((nl >> 3) & 1);
    out[o] = n < Cout ? planes[(int64_t)pl * Cout * taps * Cin + ((int64_t)n * taps + tap) * Cin + chunk * 16 + 8 * c + e]
                      : (_Float16)0.0f;
}

extern "C" int al3d_pack_f16x3_dma(const void* planes_f16x2, int Cout, int taps, int Cin, void* out_image, void* stream)
{
    AL3D_REQUIRE(planes_f16x2 && out_image, "al3d_pack_f16x3_dma: null pointer");
    const int64_t count = al3d_pack_f16x3_bstream_elems(Cout, taps, Cin);     // same size: whole 128-row blocks
    AL3D_REQUIRE(count > 0, "al3d_pack_f16x3_dma: needs Cin %% 16 == 0 (got Cout %d, taps %d, Cin %d)", Cout, taps, Cin);
    hipLaunchKernelGGL(pack_dma_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)planes_f16x2, Cout, taps, Cin, (_Float16*)out_image, count);
    AL3D_CHECK_LAUNCH("pack_dma_kernel");
    return AL3D_OK;
}

static int dma_stages()
{
    static int ns = 0;
    if (!ns) {
        const char* e = getenv("AL3D_DMA_STAGES");
        ns = e ? atoi(e) : 3;                          // 3 stages = 48 KB: three workgroups per CU (measured best)
        if (ns != 3 && ns != 4 && ns != 5) ns = 3;
    }
    return ns;
}

static int dma_pipe()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AL3D_DMA_PIPE");
        v = e ? atoi(e) != 0 : 1;
    }
    return v;
}

template <int MODE> static void launch_dma(const ConvF3Params& p, dim3 grid, hipStream_t s)
{
    // epilogue form: 16-byte stores through an LDS transposition need Cout, ldc, coff multiples of 4 and an aligned base
    static int direct = -1;                            // AL3D_DMA_EPI=direct: untransposed dword stores everywhere, for A/B
    if (direct < 0) { const char* e = getenv("AL3D_DMA_EPI"); direct = e && e[0] == 'd'; }
    const bool vec_ok = p.Cout % 4 == 0 && p.ldc % 4 == 0 && p.coff % 4 == 0 && ((uintptr_t)p.out & 15) == 0;
    if (dma_pipe() && dma_stages() == 3) {             // the shipped shape: all row / pixel format combinations
        const int io = p.io | ((p.io & 2) == 0 && (direct || !vec_ok) ? 4 : 0);
        switch (io) {
        case 0: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 0>), grid, dim3(256), 0, s, p); break;
        case 1: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 1>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 2>), grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 3>), grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 4>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 3, 5>), grid, dim3(256), 0, s, p); break;
        }
        return;
    }
    if (dma_pipe()) {
        if (dma_stages() == 5) hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 5, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv2d_f16x3_dma2_kernel<MODE, 4, 4>), grid, dim3(256), 0, s, p);
        return;
    }
    switch (dma_stages()) {
    case 4: hipLaunchKernelGGL((conv2d_f16x3_dma_kernel<MODE, 4>), grid, dim3(256), 0, s, p); break;
    case 5: hipLaunchKernelGGL((conv2d_f16x3_dma_kernel<MODE, 5>), grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((conv2d_f16x3_dma_kernel<MODE, 3>), grid, dim3(256), 0, s, p); break;
    }
}

// gap_part may be null (no fused GAP); otherwise [B][gap_parts][ldc] with gap_parts >= al3d_gap_parts_count
extern "C" int al3d_conv2d_nhwc_f16x3_dma(const float* in, const void* wgt_image, const float* scale,
                                          const float* shift, float* out, int B, int H, int W, int Cin,
                                          int Cout, int ksize, int stride, int pad, int ldc, int coff,
                                          int relu, float* gap_part, int gap_parts, int io, void* stream)
{
    ConvF3Params p;
    p.gap = gap_part; p.gap_parts = gap_parts; p.io = io;
    AL3D_REQUIRE(io >= 0 && io < 4 && (!(io & 2) || (Cout % 8 == 0 && ldc % 8 == 0 && coff % 8 == 0)),
                 "al3d_conv2d_nhwc_f16x3_dma: bad io flags (pair output needs Cout, ldc, coff multiples of 8)");
    AL3D_REQUIRE(io == 0 || (dma_pipe() && dma_stages() == 3), "al3d_conv2d_nhwc_f16x3_dma: pair pixels need the shipped kernel shape (AL3D_DMA_PIPE=1, 3 stages)");
    p.in = in; p.wgt = (const _Float16*)wgt_image; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ldc = ldc; p.coff = coff; p.relu = relu;
    AL3D_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0, "al3d_conv2d_nhwc_f16x3_dma: bad geometry");
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    AL3D_REQUIRE(p.OH >= 1 && p.OW >= 1, "al3d_conv2d_nhwc_f16x3_dma: empty output");
    p.plane = 0;
    int rc = convf3_check(p, "al3d_conv2d_nhwc_f16x3_dma");
    if (rc) return rc;
    AL3D_REQUIRE((int64_t)B * H * W * Cin < ((int64_t)1 << 31), "al3d_conv2d_nhwc_f16x3_dma: input above 2^31 elements");
    AL3D_REQUIRE(!gap_part || gap_parts >= al3d_gap_parts_count(p.OH, p.OW, 0),
                 "al3d_conv2d_nhwc_f16x3_dma: gap_parts must be at least al3d_gap_parts_count(OH, OW, 0)");
    p.tiles_x = (int)al3d_cdiv(p.OW, F3_TW);
    p.tiles_y = (int)al3d_cdiv(p.OH, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    launch_dma<0>(p, dim3(f3_grid(p)), (hipStream_t)stream);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_dma_kernel<conv>");
    return AL3D_OK;
}

extern "C" int al3d_deconv2x2_nhwc_f16x3_dma(const float* in, const void* wgt_image, const float* scale,
                                             const float* shift, float* out, int B, int H, int W, int Cin,
                                             int Cout, int ldc, int coff, int relu, float* gap_part, int gap_parts,
                                             int io, void* stream)
{
    ConvF3Params p;
    p.gap = gap_part; p.gap_parts = gap_parts; p.io = io;
    AL3D_REQUIRE(io >= 0 && io < 4 && (!(io & 2) || (Cout % 8 == 0 && ldc % 8 == 0 && coff % 8 == 0)),
                 "al3d_deconv2x2_nhwc_f16x3_dma: bad io flags (pair output needs Cout, ldc, coff multiples of 8)");
    AL3D_REQUIRE(io == 0 || (dma_pipe() && dma_stages() == 3), "al3d_deconv2x2_nhwc_f16x3_dma: pair pixels need the shipped kernel shape (AL3D_DMA_PIPE=1, 3 stages)");
    p.in = in; p.wgt = (const _Float16*)wgt_image; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ksize = 2; p.stride = 2; p.pad = 0; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.OH = 2 * H; p.OW = 2 * W;
    p.plane = 0;
    int rc = convf3_check(p, "al3d_deconv2x2_nhwc_f16x3_dma");
    if (rc) return rc;
    AL3D_REQUIRE((int64_t)B * H * W * Cin < ((int64_t)1 << 31), "al3d_deconv2x2_nhwc_f16x3_dma: input above 2^31 elements");
    AL3D_REQUIRE(!gap_part || gap_parts >= al3d_gap_parts_count(p.OH, p.OW, 1),
                 "al3d_deconv2x2_nhwc_f16x3_dma: gap_parts must be at least al3d_gap_parts_count(OH, OW, 1)");
    p.tiles_x = (int)al3d_cdiv(W, F3_TW);
    p.tiles_y = (int)al3d_cdiv(H, F3_TH);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = (int)al3d_cdiv(Cout, F3_BN);
    launch_dma<1>(p, dim3(f3_grid(p), 1, 4), (hipStream_t)stream);
    AL3D_CHECK_LAUNCH("conv2d_f16x3_dma_kernel<deconv>");
    return AL3D_OK;
}
