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
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                int oy = y, ox = x;
                if (MODE == 1) { oy = 2 * y + (tap0 >> 1); ox = 2 * x + (tap0 & 1); }
                p.out[(((int64_t)b * p.OH + oy) * p.OW + ox) * p.ldc + p.coff + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1
// The generic kernel above re-stages (and re-splits) every input pixel once per filter tap, and
// its LDS write traffic (activations + weights, 24 KB per step) is what bounds it.  For the
// twelve 3x3 stride-1 layers of the neck the input halo of the output tile is staged ONCE per
// 16-channel chunk -- 6x34 pixels for a 4x32 output tile -- and the nine taps read their A
// fragments from it at shifted rows; only the weight tile changes per tap.  LDS writes per
// step drop from 24 KB to ~14 KB and the split arithmetic by 6x.
// Output tile = 4 rows x 32 columns: lane r of an M-tile is column r, so the 32 lanes of a
// fragment read 32 consecutive halo rows (conflict-free ds_read_b128 at a 48-byte pitch).
#define H3_TH 4
#define H3_TW 32
#define H3_HH (H3_TH + 2)
#define H3_HW (H3_TW + 2)
#define H3_HP (H3_HH * H3_HW)          // 204 halo pixels

// Software pipeline: a tap that starts with its twelve ds_read_b128 right after the barrier has
// nothing to cover their latency (measured: +4 % when removed).  The fragments of tap t+1 are read
// into a second register set while the 24 MFMAs of tap t run, so a step opens with MFMAs whose
// operands are already in registers:
//   * weights live in THREE LDS buffers (tap t, t+1 readable; t+2 being written) -- 32-byte rows
//     with the 16-byte halves XOR-swizzled by bit 3 of the row (conflict-free ds_read_b128 /
//     ds_write_b128 without the 16-byte pad), so three buffers cost what two padded ones did;
//   * still one barrier per tap: it publishes B(t+2) and retires B(t) (whose fragments were
//     read during tap t-1);
//   * the next chunk's halo is stored during tap 8 (its fragments were read during tap 7) and
//     published by tap 8's barrier; only tap 0 of a chunk reads its fragments unprefetched.
#define H3_BROW 32                     // bytes per weight row in LDS (16 bf16, swizzled halves)
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x6_halo_kernel(Conv6Params p)
{
    __shared__ __attribute__((aligned(16))) unsigned char Ah[3][H3_HP * C6_LDB];          // 29.4 KB
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3][3][C6_BN * H3_BROW];     // 36.9 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fh = lane >> 5;
    int tile = blockIdx.x;
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = blockIdx.y * C6_BN;
    const int y0 = ty_ * H3_TH - 1, x0 = tx_ * H3_TW - 1;       // image coords of halo (0,0)
    const int nchunks = p.Cin / C6_BK;
    const int bq = tid & 1, br = tid >> 1;

    float4 rh[4];
    uint4 rb[3];
    auto load_halo = [&](int chunk) {
        const int c0 = chunk * C6_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int hp = piece >> 2, q = piece & 3;
            const int iy = y0 + hp / H3_HW, ix = x0 + hp % H3_HW;
            const bool ok = hp < H3_HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            rh[i] = ok ? *reinterpret_cast<const float4*>(
                             p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + c0 + 4 * q)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = tid + 256 * i;
            const int hp = piece >> 2, q = piece & 3;
            if (hp >= H3_HP) continue;
            const float v[4] = {rh[i].x, rh[i].y, rh[i].z, rh[i].w};
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 a, bb, c; split3(v[e], a, bb, c); h[e] = a; m[e] = bb; l[e] = c; }
            const int off = hp * C6_LDB + 8 * q;
            *reinterpret_cast<bf16x4*>(&Ah[0][off]) = h;
            *reinterpret_cast<bf16x4*>(&Ah[1][off]) = m;
            *reinterpret_cast<bf16x4*>(&Ah[2][off]) = l;
        }
    };
    const int nb = n0 + br;
    const bool nb_ok = nb < p.Cout;
    const __bf16* wrow = p.wgt + (int64_t)nb * 9 * p.Cin + 8 * bq;     // this thread's weight row
    auto load_b = [&](int chunk, int tap) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            rb[pl] = nb_ok ? *reinterpret_cast<const uint4*>(wrow + pl * p.plane + tap * p.Cin + chunk * C6_BK)
                           : make_uint4(0u, 0u, 0u, 0u);
    };
    const int bw_off = br * H3_BROW + 16 * (bq ^ ((br >> 3) & 1));
    auto store_b = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(&Bs[buf][pl][bw_off]) = rb[pl];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int total = 9 * nchunks;                    // steps; step s = chunk * 9 + tap, B(s) lives in buffer tap % 3
    load_halo(0);
    load_b(0, 0);
    store_halo();
    store_b(0);
    load_b(0, 1);
    store_b(1);
    __syncthreads();
    const int a_off = ((2 * wm) * H3_HW + fr) * C6_LDB + 16 * fh;
    const int b_off = (wn * 64 + fr) * H3_BROW + 16 * (fh ^ ((fr >> 3) & 1));   // +32 rows keeps bit 3
    bf16x8 fa[2][3][2], fb[2][3][2];                  // [set][plane][tile]
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {           // compile-time taps: immediate LDS offsets, static sets
            const int cs = tap & 1, ns = cs ^ 1;
            // ---- weights two steps ahead (registers now, LDS at the end of the step)
            const bool has2 = chunk * 9 + tap + 2 < total;
            if (tap < 7) load_b(chunk, tap + 2);
            else if (more) load_b(chunk + 1, tap - 7);
            if (tap == 0 && more) load_halo(chunk + 1);             // lands during the 9 taps
            if (tap == 0) {                                         // first tap of a chunk: fragments not prefetched
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[0][pl][t] = *reinterpret_cast<const bf16x8*>(&Ah[pl][a_off + (t * H3_HW) * C6_LDB]);
                        fb[0][pl][t] = *reinterpret_cast<const bf16x8*>(&Bs[0][pl][b_off + t * 32 * H3_BROW]);
                    }
            }
            if (tap < 8) {                                          // fragments of the next tap
                const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[ns][pl][t] = *reinterpret_cast<const bf16x8*>(
                            &Ah[pl][a_off + ((t + ky) * H3_HW + kx) * C6_LDB]);
                        fb[ns][pl][t] = *reinterpret_cast<const bf16x8*>(
                            &Bs[(tap + 1) % 3][pl][b_off + t * 32 * H3_BROW]);
                    }
            }
            if (tap == 8 && more) store_halo();                     // halo(chunk) was last read during tap 7
            {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cs][PA[t]][i], fb[cs][PB[t]][j], acc[i][j], 0, 0, 0);
            }
            if (has2) store_b((tap + 2) % 3);
            __syncthreads();
        }
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = ty_ * H3_TH + 2 * wm + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = tx_ * H3_TW + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (y >= p.OH || x >= p.OW) continue;
                float v = acc[i][j][r] * sc + sh;
                if (p.relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
                p.out[(((int64_t)b * p.OH + y) * p.OW + x) * p.ldc + p.coff + n] = v;
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

// inverse of the split: f32 = (x1 + x2) + x3, exact
__global__ void merge_planes_kernel(const __bf16* __restrict__ pl, int64_t count, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = ((float)pl[i] + (float)pl[count + i]) + (float)pl[2 * count + i];
}

extern "C" int al3d_merge_bf16x3(const void* planes_bf16x3, int64_t count, float* out, void* stream)
{
    AL3D_REQUIRE(count >= 0, "al3d_merge_bf16x3: bad count");
    if (count == 0) return AL3D_OK;
    AL3D_REQUIRE(planes_bf16x3 && out, "al3d_merge_bf16x3: null pointer");
    hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)planes_bf16x3, count, out);
    AL3D_CHECK_LAUNCH("merge_planes_kernel");
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
    if (ksize == 3 && stride == 1 && pad == 1) {     // halo-staged fast path
        p.tiles_x = (int)al3d_cdiv(p.OW, H3_TW);
        p.tiles_y = (int)al3d_cdiv(p.OH, H3_TH);
        dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)al3d_cdiv(Cout, C6_BN), 1);
        hipLaunchKernelGGL(conv3x3_bf16x6_halo_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
        AL3D_CHECK_LAUNCH("conv3x3_bf16x6_halo_kernel");
        return AL3D_OK;
    }
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
