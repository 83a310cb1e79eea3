// 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) in f16x3 arithmetic (conv2d_f16x3.hip): 16 multiplies per
// 2 x 2 output tile and (input, output) channel pair instead of 36, i.e. 2.25 x fewer matrix-core products on the
// eleven stride-1 3x3 layers of the SECOND neck (det3d/models/necks/rpn.py:66-113 -- 58 of its 67.6 GFLOP per frame).
//
//   V = B^T d B   (4 x 4 input patch d, stride 2; entries are +-sums of four pixels: fp32 adds)
//   U = G g G^T   (computed once per layer in float64 on the host, rounded to fp32, then split like any f16x3 weight)
//   M_p = sum_cin V_p U_p  for the 16 positions p of the 4 x 4 transformed patch  (the matrix-core part)
//   Y = A^T M A   (2 x 2 outputs; fp32 adds), then BN scale / shift / ReLU
//
// Workgroup = 256 threads, output tile 16 x 16 pixels (64 Winograd tiles) x 64 output channels, input channels in
// chunks of 16.  Per chunk the raw fp32 halo (18 x 18 pixels x 16 channels) arrives in LDS by DMA (ring of three stages,
// one barrier per chunk).  Wave w owns the four positions of transformed ROW w, and every lane computes its own A
// fragments in registers: lane (tile, k half) reads the two patch rows that row w combines (8 pixels x 8 channels), forms
// t = d[r1] +- d[r2], the four column combinations, and splits them (xh, xl' = residual x 2^11) -- no transformed image
// in LDS, no barrier between transform and products.  2 x 2 MFMA tiles of 32 x 32 per position, B fragments streamed
// from L2 in fragment order (al3d_pack_f16x3_wino), three products per MAC into one fp32 accumulator (xl' wd, xh wl,
// xh wh).  256 accumulator registers per lane: one workgroup per CU.
// The output transform runs along the row inside a wave's registers and across the four waves through LDS.
//
// Numerics: measured error against float64 stays at the fp32-input kernel's level (tests/test_dense_gpu.py); the
// kernel is NOT bit-identical to the direct kernels (another summation tree).
// Status (round 3): at parity with the direct streamed kernel, not ahead -- with 256 accumulator registers
// only one wave fits a SIMD, and the input transform + split (~8 vector instructions per MFMA) is not hidden behind the
// products by the compiler's schedule (a software-pipelined variant with sched_group_barrier measured the same); it
// stays opt-in (AL3D_DENSE=wino).  gfx950 only.
#include "al3d_common.h"
#include "sp_rows.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define WN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define WN_IC(v) std::integral_constant<int, v>{}
#define WN_T 16                         // output tile edge
#define WN_HW (WN_T + 2)                // halo edge
#define WN_HP (WN_HW * WN_HW)           // 324 halo pixels
#define WN_STAGE 24576                  // bytes per halo stage: 24 DMA instructions of 16 pixels x 64 B (324 pixels used)
#define WN_NS 3
#define WN_XP 36                        // floats per (row a, j, tile) line of the output exchange: 32 channels + pad

typedef float wn_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void wn_lds_void;
typedef const __attribute__((address_space(1))) void wn_gbl_void;

struct WinoParams {
    const float* in;        // [B, H, W, Cin] f32
    const _Float16* wgt;    // [2 planes][Cout/32][Cin/16][16 positions][64 lanes][8] f16 (al3d_pack_f16x3_wino)
    const float* scale;     // BN scale * 2^-s
    const float* shift;
    float* out;             // [B, H, W, ldc]
    int B, H, W, Cin, Cout, ldc, coff, relu;
    int tiles_x, tiles_y, ntiles, nblocks;
};

__device__ __attribute__((aligned(64))) float g_wn_zero[16];

__device__ __forceinline__ void wn_split(float x, _Float16& h, _Float16& l)
{
    h = (_Float16)x;
    l = (_Float16)__builtin_fmaf((float)h, -2048.0f, x * 2048.0f);
}

// same XCD-aware (pixel tile, channel block) order as conv2d_f16x3.hip
__device__ __forceinline__ bool wn_tile_of_block(const WinoParams& p, int& tile, int& nblk)
{
    const int id = blockIdx.x, span = 8 * p.nblocks;
    const int grp = id / span, rem = id - grp * span;
    nblk = rem >> 3;
    tile = grp * 8 + (rem & 7);
    return tile < p.ntiles;
}

// The halo's 16-byte chunk q of pixel (row, col) sits at position q ^ ((col >> 2) & 3) of the pixel's 64 bytes (applied
// on the source side of the DMA): the eight tiles of a tile row then read eight different bank groups.
__device__ __forceinline__ int wn_swz(int col) { return (col >> 2) & 3; }

template <int IO, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_f16x3_wino_kernel(WinoParams p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WN_NS * WN_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    int tile, nblk;
    if (!wn_tile_of_block(p, tile, nblk)) return;
    const int tx_ = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y; tile /= p.tiles_y;
    const int b = tile;
    const int n0 = nblk * 64;
    const int y0 = ty_ * WN_T - 1, x0 = tx_ * WN_T - 1;
    const int nchunks = p.Cin >> 4;
    const unsigned smem_base = (unsigned)(size_t)(wn_lds_void*)smem;

    // ---- halo by LDS-DMA: instruction k of wave w covers halo pixels 16 (6 w + k) .. + 15; lane (px = l >> 2, pos = l & 3)
    // fetches chunk pos ^ swz(col).  Addresses depend on the chunk only through a constant stride.
    const char* hsrc[6];
    unsigned hinc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int hp = 16 * (6 * wave + k) + (lane >> 2);
        const int row = hp / WN_HW, col = hp - row * WN_HW;
        const int iy = y0 + row, ix = x0 + col;
        const int q = (lane & 3) ^ wn_swz(col);
        const bool ok = hp < WN_HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        hsrc[k] = reinterpret_cast<const char*>(ok ? p.in + (((int64_t)b * p.H + iy) * p.W + ix) * p.Cin + 4 * q : g_wn_zero + 4 * q);
        hinc[k] = ok ? 64u : 0u;
    }
    auto issue_halo = [&](int chunk) {
        const int cc = chunk < nchunks ? chunk : nchunks - 1;          // past the end: re-fetch the last chunk (6 DMAs, always)
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + (chunk % WN_NS) * WN_STAGE + wave * 6144);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            __builtin_amdgcn_global_load_lds((wn_gbl_void*)(hsrc[k] + (size_t)cc * hinc[k]), (wn_lds_void*)(size_t)(dst + k * 1024), 16, 0, 0);
    };

    // ---- this wave's weight stream: positions 4 w .. 4 w + 3 (transformed row a = w), both 32-channel tiles of the block.
    // Loaded by inline asm (the compiler would hoist plain loads to the top of the loop and then drain vmcnt(0), halo DMAs
    // included, before the first product) into the registers the previous chunk's products have just released.
    const int NT = p.Cout >> 5;
    const _Float16* bsrc[2][2];                       // [plane][n tile]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bsrc[pl][j] = p.wgt + ((((int64_t)pl * NT + (n0 >> 5) + j) * nchunks) * 16 + wave * 4) * 512 + lane * 8;
    f16x8 fb[4][2][2];                                // [position in the row][plane][n tile]
    auto load_b = [&](int chunk, int c, f16x8 (&f)[2][2]) {
        const int cc = chunk < nchunks ? chunk : nchunks - 1;
        const int64_t o = ((int64_t)cc * 16 + c) * 1024;        // bytes
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                     : "=&v"(f[0][0]), "=&v"(f[0][1]), "=&v"(f[1][0]), "=&v"(f[1][1])
                     : "v"(reinterpret_cast<const char*>(bsrc[0][0]) + o), "v"(reinterpret_cast<const char*>(bsrc[0][1]) + o),
                       "v"(reinterpret_cast<const char*>(bsrc[1][0]) + o), "v"(reinterpret_cast<const char*>(bsrc[1][1]) + o)
                     : "memory");
    };
    // Position c's weights are waited for where its first product is issued (not all four at the top of the chunk): in
    // issue order behind them sit the later positions' loads (4 each) and the halo DMAs of chunk + 2 (6), so "all but the
    // youngest 18 / 14 / 10 / 6" covers positions 0 / 1 / 2 / 3 -- the last position's request, issued at the very end of
    // the previous chunk, gets three quarters of this chunk's first tile as cover.
    auto wait_b_pos = [&](int c, f16x8 (&f)[2][2]) {
        if (c == 0) asm volatile("s_waitcnt vmcnt(18)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]) :: "memory");
        else if (c == 1) asm volatile("s_waitcnt vmcnt(14)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]) :: "memory");
        else if (c == 2) asm volatile("s_waitcnt vmcnt(10)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]) :: "memory");
    };

    // ---- this lane's patch rows.  Transformed row a = wave:  t = d[r1] + sg d[r2]  with (r1, r2, sg) =
    // (0,2,-), (1,2,+), (2,1,-), (1,3,-);  then V[a][c] = t0 - t2, t1 + t2, t2 - t1, t1 - t3.
    const int r1 = wave == 0 ? 0 : wave == 2 ? 2 : 1, r2 = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const float sg = wave == 1 ? 1.0f : -1.0f;
    // byte offsets of this lane's reads in a stage: (tile 0, row r1) per (pixel c, 16-byte half); tile 1 is eight halo
    // rows further (+ 9216 B), row r2 a wave-uniform distance away
    unsigned ao[4][2];
    {
        const int ty = fr >> 3, tx = fr & 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 2 * tx + c, hp = (2 * ty + r1) * WN_HW + col;
#pragma unroll
            for (int e = 0; e < 2; ++e) ao[c][e] = (unsigned)(hp * 64 + (((2 * fh + e) ^ wn_swz(col)) << 4));
        }
    }
    const int rdelta = __builtin_amdgcn_readfirstlane((r2 - r1) * WN_HW * 64);

    f32x16 acc[4][2][2];                              // [position in the row][m tile][n tile]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    // The patch rows of a tile are read and transformed in two halves (the fragment's two 16-byte pieces = channels
    // 8 h .. 8 h + 3 and 8 h + 4 .. 8 h + 7): eight 16-byte reads and 32 live registers at a time instead of sixteen / 64.
    wn_f32x4 d[2][4];                                 // [row r1 / r2][pixel c]
    auto patch_read = [&](unsigned sb, int e) {       // sb = stage base + tile offset (0 / 9216)
        const unsigned s1 = sb + rdelta;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                     "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
                     : "=&v"(d[0][0]), "=&v"(d[0][1]), "=&v"(d[0][2]), "=&v"(d[0][3]), "=&v"(d[1][0]), "=&v"(d[1][1]), "=&v"(d[1][2]),
                       "=&v"(d[1][3])
                     : "v"(sb + ao[0][e]), "v"(sb + ao[1][e]), "v"(sb + ao[2][e]), "v"(sb + ao[3][e]),
                       "v"(s1 + ao[0][e]), "v"(s1 + ao[1][e]), "v"(s1 + ao[2][e]), "v"(s1 + ao[3][e])
                     : "memory");
    };
    auto patch_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]), "+v"(d[1][3])
                     :: "memory");
    };
    // d (half e) -> elements 4 e .. 4 e + 3 of the four A fragment pairs of the row's positions
    auto transform = [&](f16x8 (&ah)[4], f16x8 (&al)[4], int e) {
        float t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[c][i] = __builtin_fmaf(sg, d[1][c][i], d[0][c][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v0 = t[0][i] - t[2][i], v1 = t[1][i] + t[2][i], v2 = t[2][i] - t[1][i], v3 = t[1][i] - t[3][i];
            _Float16 hh, ll;
            wn_split(v0, hh, ll); ah[0][4 * e + i] = hh; al[0][4 * e + i] = ll;
            wn_split(v1, hh, ll); ah[1][4 * e + i] = hh; al[1][4 * e + i] = ll;
            wn_split(v2, hh, ll); ah[2][4 * e + i] = hh; al[2][4 * e + i] = ll;
            wn_split(v3, hh, ll); ah[3][4 * e + i] = hh; al[3][4 * e + i] = ll;
        }
    };
    auto make_a = [&](unsigned sb, f16x8 (&ah)[4], f16x8 (&al)[4]) {     // a whole tile's fragments (two halves)
        if constexpr (ABL == 2 || ABL == 5) {              // dev ablation: no patch reads, no transform
#pragma unroll
            for (int c = 0; c < 4; ++c) { ah[c] = fb[c][0][0]; al[c] = fb[c][1][0]; }
            return;
        }
        if constexpr (ABL != 3) { patch_read(sb, 0); patch_wait(); }
        transform(ah, al, 0);
        if constexpr (ABL != 3) { patch_read(sb, 1); patch_wait(); }
        transform(ah, al, 1);
    };
    // the 24 products of one (chunk, m tile); after_pos(c) runs when position c's six are issued
    auto products = [&](auto m_, const f16x8 (&ah)[4], const f16x8 (&al)[4], auto&& after_pos) {
        constexpr int m = decltype(m_)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (m == 0) wait_b_pos(c, fb[c]);
            f16x8 wd[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) wd[j] = fb[c][0][j] * (_Float16)0.00048828125f;
            if constexpr (ABL == 1) {                    // dev ablation: no products (operands kept alive)
                asm volatile("" :: "v"(al[c]), "v"(ah[c]), "v"(wd[0]), "v"(wd[1]), "v"(fb[c][1][0]), "v"(fb[c][1][1]), "v"(fb[c][0][0]), "v"(fb[c][0][1]));
            } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[c][m][j] = WN_MFMA(al[c], wd[j], acc[c][m][j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[c][m][j] = WN_MFMA(ah[c], fb[c][1][j], acc[c][m][j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[c][m][j] = WN_MFMA(ah[c], fb[c][0][j], acc[c][m][j]);
            }
            after_pos(c);
        }
    };
    f16x8 ah0[4], al0[4];
    {
        // ---- plain loop: per chunk one barrier; per tile patch reads -> transform -> 24 products.  The next chunk's weights
        // are requested position by position as the second tile's products release their registers; the halo of chunk + 2 is
        // requested behind them, so waiting for the weights ("all but the youngest 6") leaves that DMA in flight.
        issue_halo(0);
        load_b(0, 0, fb[0]); load_b(0, 1, fb[1]); load_b(0, 2, fb[2]); load_b(0, 3, fb[3]);
        issue_halo(1);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            asm volatile("s_waitcnt vmcnt(22)" ::: "memory");           // halo(chunk): behind it this chunk's 16 weight loads and halo(chunk + 1)
            __builtin_amdgcn_s_barrier();              // everyone's share landed; every wave is past chunk - 1's LDS reads
            const unsigned sb = smem_base + (chunk % WN_NS) * WN_STAGE;
            make_a(sb, ah0, al0);
            products(WN_IC(0), ah0, al0, [](int) {});
            make_a(sb + 8 * WN_HW * 64, ah0, al0);
            products(WN_IC(1), ah0, al0, [&](int c) {
                if (c == 0) load_b(chunk + 1, 0, fb[0]);
                else if (c == 1) load_b(chunk + 1, 1, fb[1]);
                else if (c == 2) load_b(chunk + 1, 2, fb[2]);
                else load_b(chunk + 1, 3, fb[3]);
            });
            issue_halo(chunk + 2);                     // -> the stage chunk - 1 used (free since this chunk's barrier)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's dummy requests must not outlive the workgroup's LDS

    // ---- output transform.  Wave w holds row a = w of M (positions (w, 0..3)):  P_w[j] = sum_c M[w][c] A^T[j][c]
    // in registers (A^T = [[1,1,1,0],[0,1,-1,-1]]), then Y[i][j] = sum_a A^T[i][a] P_a[j] across the waves through LDS,
    // one 32-channel half at a time: X[a][j][tile][32 + pad] floats = 72 KB in the halo ring's storage.
    if constexpr (ABL == 4 || ABL == 5) {                // dev ablation: no output transform / stores (one word keeps the products alive)
        float keep = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[q][i][j][r];
        if (keep == 123.456f) p.out[0] = keep;
        return;
    }
    float* X = reinterpret_cast<float*>(smem);
    static_assert(4 * 2 * 64 * WN_XP * 4 <= WN_NS * WN_STAGE, "exchange must fit the halo ring");
#pragma unroll
    for (int j_n = 0; j_n < 2; ++j_n) {
        __syncthreads();                               // the ring (or the previous half's exchange) is no longer read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float m0 = acc[0][i][j_n][r], m1 = acc[1][i][j_n][r], m2 = acc[2][i][j_n][r], m3 = acc[3][i][j_n][r];
                X[((wave * 2 + 0) * 64 + tl) * WN_XP + fr] = (m0 + m1) + m2;
                X[((wave * 2 + 1) * 64 + tl) * WN_XP + fr] = (m1 - m2) - m3;
            }
        __syncthreads();
        // 256 output pixels x 32 channels of this half: a lane takes 8 consecutive channels of a pixel
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int item = tid + 256 * pass;         // 1,024 items = 256 pixels x 4 channel groups
            const int g = item & 3, px = item >> 2;
            const int oy = px >> 4, ox = px & 15;
            const int tl = (oy >> 1) * 8 + (ox >> 1), i = oy & 1, j = ox & 1;
            const int y = ty_ * WN_T + oy, x = tx_ * WN_T + ox;
            const int n = n0 + 32 * j_n + 8 * g;
            if (y >= p.H || x >= p.W || n >= p.Cout) continue;
            float v[8];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const float* xp = X + ((0 * 2 + j) * 64 + tl) * WN_XP + 8 * g + 4 * hlf;
                const float4 p0 = *reinterpret_cast<const float4*>(xp);
                const float4 p1 = *reinterpret_cast<const float4*>(xp + 2 * 64 * WN_XP);
                const float4 p2 = *reinterpret_cast<const float4*>(xp + 4 * 64 * WN_XP);
                const float4 p3 = *reinterpret_cast<const float4*>(xp + 6 * 64 * WN_XP);
                float4 yv;
                if (i == 0) yv = make_float4((p0.x + p1.x) + p2.x, (p0.y + p1.y) + p2.y, (p0.z + p1.z) + p2.z, (p0.w + p1.w) + p2.w);
                else yv = make_float4((p1.x - p2.x) - p3.x, (p1.y - p2.y) - p3.y, (p1.z - p2.z) - p3.z, (p1.w - p2.w) - p3.w);
                v[4 * hlf + 0] = yv.x; v[4 * hlf + 1] = yv.y; v[4 * hlf + 2] = yv.z; v[4 * hlf + 3] = yv.w;
            }
            const float4 s0 = *reinterpret_cast<const float4*>(p.scale + n), s1 = *reinterpret_cast<const float4*>(p.scale + n + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.shift) {
                const float4 t0 = *reinterpret_cast<const float4*>(p.shift + n), t1 = *reinterpret_cast<const float4*>(p.shift + n + 4);
                sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = v[e] * sc[e] + sh[e];
                if (p.relu) v[e] = v[e] <= 0.f ? 0.f : v[e];           // NaN propagates, like torch.relu
            }
            float* o = p.out + (((int64_t)b * p.H + y) * p.W + x) * p.ldc + p.coff + n;
            if constexpr ((IO & SP_IO_OUT_PAIR) != 0) {
                uint4 hi, lo;
                sp_split8(v, hi, lo);
                *reinterpret_cast<uint4*>(o) = hi;
                *reinterpret_cast<uint4*>(o + 4) = lo;
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
}

// planes [2][Cout][16][Cin] f16 (al3d_split_f16x3 of U = G g G^T, position p = a * 4 + c) ->
// [2][Cout/32][Cin/16][16][64 lanes][8]: lane (r, h) of fragment (n tile, chunk, position) holds U[32 nt + r][16 chunk + 8 h + j]
__global__ void pack_wino_kernel(const _Float16* __restrict__ planes, int Cout, int Cin, _Float16* __restrict__ out, int64_t count)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= count) return;
    const int nch = Cin >> 4;
    int64_t r = o;
    const int j = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int pos = r % 16; r /= 16;
    const int chunk = r % nch; r /= nch;
    const int nt = r % (Cout >> 5); r /= (Cout >> 5);
    const int pl = (int)r;
    const int n = nt * 32 + (lane & 31), cin = chunk * 16 + 8 * (lane >> 5) + j;
    out[o] = planes[(((int64_t)pl * Cout + n) * 16 + pos) * Cin + cin];
}

extern "C" int al3d_pack_f16x3_wino(const void* planes_f16x2, int Cout, int Cin, void* out_frag, void* stream)
{
    AL3D_REQUIRE(planes_f16x2 && out_frag, "al3d_pack_f16x3_wino: null pointer");
    AL3D_REQUIRE(Cout >= 64 && Cout % 64 == 0 && Cin >= 16 && Cin % 16 == 0, "al3d_pack_f16x3_wino: Cout %% 64, Cin %% 16 (got %d, %d)", Cout, Cin);
    const int64_t count = (int64_t)2 * Cout * 16 * Cin;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)al3d_cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)planes_f16x2, Cout, Cin, (_Float16*)out_frag, count);
    AL3D_CHECK_LAUNCH("pack_wino_kernel");
    return AL3D_OK;
}

extern "C" int al3d_conv3x3_nhwc_f16x3_wino(const float* in, const void* wgt_wino, const float* scale, const float* shift,
                                            float* out, int B, int H, int W, int Cin, int Cout, int ldc, int coff, int relu,
                                            int io, void* stream)
{
    AL3D_REQUIRE(in && wgt_wino && scale && out, "al3d_conv3x3_nhwc_f16x3_wino: null pointer (scale carries the weight exponent and is required)");
    AL3D_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 16 && Cin % 16 == 0 && Cout >= 64 && Cout % 64 == 0,
                 "al3d_conv3x3_nhwc_f16x3_wino: Cin %% 16, Cout %% 64 (got %d, %d)", Cin, Cout);
    AL3D_REQUIRE(io == 0 || io == SP_IO_OUT_PAIR, "al3d_conv3x3_nhwc_f16x3_wino: io = 0 or 2 (f32 pixels in)");
    AL3D_REQUIRE(coff >= 0 && coff + Cout <= ldc && ldc % 8 == 0 && coff % 8 == 0, "al3d_conv3x3_nhwc_f16x3_wino: channel window");
    AL3D_REQUIRE((((uintptr_t)in | (uintptr_t)wgt_wino | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
                 "al3d_conv3x3_nhwc_f16x3_wino: pointers must be 16-byte aligned");
    WinoParams p;
    p.in = in; p.wgt = (const _Float16*)wgt_wino; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ldc = ldc; p.coff = coff; p.relu = relu;
    p.tiles_x = (int)al3d_cdiv(W, WN_T); p.tiles_y = (int)al3d_cdiv(H, WN_T);
    p.ntiles = p.tiles_x * p.tiles_y * B; p.nblocks = Cout / 64;
    const dim3 grid((unsigned)(al3d_cdiv(p.ntiles, 8) * 8 * p.nblocks));
#ifdef AL3D_WINO_ABLATE
    hipLaunchKernelGGL((conv3x3_f16x3_wino_kernel<0, AL3D_WINO_ABLATE>), grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv3x3_f16x3_wino_kernel");
    return AL3D_OK;
#endif
    if (io == 0) hipLaunchKernelGGL(conv3x3_f16x3_wino_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_f16x3_wino_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("conv3x3_f16x3_wino_kernel");
    return AL3D_OK;
}
