// Sparse convolution with an LDS-DMA row gather (gfx950), f16x3 arithmetic.
//
// Why: sp_conv_wave2_kernel (spconv_wave.hip) loads every gathered row FRAGMENT-SHAPED -- lane (r, h)
// of a wave takes 32 bytes of row r straight into the registers the MFMA wants -- so one
// `global_load_dwordx4` touches 64 different (row, 16-byte) pieces and the CU's texture-address unit
// spends a tag lookup on each: its counters showed 42-51 % of the wave-cycles in s_waitcnt, 13-36 %
// MFMA-busy and a cost that did not move with hit rate or bytes.  MI355X_MICROARCH.md
// ("Projection GEMM at M = 256", x operand) measures the same thing: fragment-shaped loads cost twice
// the TA time of full-line loads.
//
// Here every gathered row is fetched as FULL 128-byte lines by `global_load_lds_dwordx4` (LDS-DMA with
// a per-lane source address): eight adjacent lanes read the eight 16-byte chunks of one row, a wave
// instruction moves 8 rows x 128 B = 1 KiB with 8 line lookups instead of 64, straight into LDS without
// passing through VGPRs.  The LDS image of a piece is lane-linear (DMA rule), so the bank swizzle is
// applied on the SOURCE side: lane (j, s) of piece i fetches chunk s ^ f(row) of row NP*j + i, and the
// fragment reads (`ds_read_b128`, two per 16-channel unit) un-swizzle with the same f -- conflict-free
// for the b128 lane groups (verified exhaustively, tools/probe_glds_swizzle.py).
//
// Structure of a workgroup: NW consumer waves, each owning R tiles of 32 output rows x all output
// channels (accumulators in registers), plus ONE producer wave that streams the weight slabs
// (pre-packed in LDS image order by al3d_sp_pack_glds_f16x3) into a ring of NB LDS buffers with the same
// DMA.  A consumer walks its ITEM list -- the (tap, channel chunk, tile) triples its tiles need, built
// once from the rulebook's per-tile tap masks -- and per item issues one DMA of the tile's 32 neighbour
// indices into a two-entry LDS ring (one item ahead of the gather that uses them) and NP row DMAs into
// a ring of P gather slots.  Those are the only VMEM operations of its main loop, always NP + 1 per
// item (items past the end gather the zero row), so the counted waits `vmcnt(NP+1)` / `vmcnt((P-1)*(NP+1))`
// retire exactly the index entry / gather slot needed next while P-1 later gathers stay in flight.
// The B fragments of a (tap, chunk) are read from LDS once and reused by the wave's R tiles; slabs are
// handed over with one raw `s_barrier` per slab (the producer waits for its DMAs before arriving).  All
// LDS reads of the main loop are inline asm with hand-placed waits: hipcc would otherwise drain every
// in-flight DMA before any LDS read it cannot prove disjoint from them.
//
// Arithmetic and summation order are EXACTLY sp_conv_wave2_kernel's (taps ascending, channel groups
// ascending, al*wd then ah*wl then ah*wh into one fp32 accumulator): outputs are bit-identical, which
// is what tests/test_detector_gpu.py::test_sparse_conv_glds_kernel_is_bit_identical checks.
#include "glds_common.h"
#include "sp_rows.h"

template <int CIN, int COUT, int NW, int R, int UPS, int P, int NB>
struct GldsCfg {
    static constexpr int KG = CIN / 16;                  // 16-channel units per tap
    static constexpr int CH = CIN >= 32 ? 32 : 16;       // channels gathered per item
    static constexpr int UA = CH / 16;                   // units per item
    static constexpr int NCC = CIN / CH;                 // channel chunks per tap
    static constexpr int TN = (COUT + 31) / 32;
    static constexpr int NROWS = TN * 32;                // weight rows per plane in the LDS image (zero rows beyond COUT)
    static constexpr int WPITCH = 32;                    // bytes per weight row: 16 f16, the two 16-byte halves swizzled
    static constexpr int UNIT_BYTES = 2 * NROWS * WPITCH;
    static constexpr int UNIT_PIECES = UNIT_BYTES / 1024;
    static constexpr int SPS = UPS / UA;                 // workgroup steps (tap, chunk) per slab
    static constexpr int LPR = CH / 4;                   // lanes (16-byte chunks) per gathered row
    static constexpr int RPP = 64 / LPR;                 // rows per DMA piece
    static constexpr int NP = 32 / RPP;                  // pieces (DMA instructions) per gather
    static constexpr int SLOT_BYTES = NP * 1024;
    static constexpr int SLAB_BYTES = UPS * UNIT_BYTES;
    static constexpr int SLAB_PIECES = UPS * UNIT_PIECES;
    static constexpr int W_BYTES = NB * SLAB_BYTES;      // ring of NB slabs: the producer runs NB-1 slabs ahead
    static constexpr int A_WAVE_BYTES = P * SLOT_BYTES;
    static constexpr int A_BYTES = NW * A_WAVE_BYTES;
    static constexpr int X_WAVE_BYTES = 2 * 256;         // two index entries (32 rows, duplicated over the wave's halves)
    static constexpr int SMEM_BYTES = W_BYTES + A_BYTES + NW * X_WAVE_BYTES;
    static constexpr int T = NW * R;                     // tiles per workgroup
    static constexpr int MAXITEMS = 27 * NCC * R;
    static constexpr int LREG = (MAXITEMS + 63) / 64;    // item list: 16 bits per item, one per lane and register
    static constexpr bool BHOLD = R > 1;                 // keep a (tap, chunk)'s B fragments in registers across the tiles
    static constexpr int EP_PITCH = 40;                  // floats (see spconv_wave.hip)
    static_assert(UNIT_BYTES % 1024 == 0, "a weight unit must be whole DMA pieces");
    static_assert(UPS % UA == 0 && KG % UA == 0, "an item must not straddle slabs");
    static_assert(A_WAVE_BYTES >= 32 * EP_PITCH * 4 && A_WAVE_BYTES >= MAXITEMS * 2,
                  "the epilogue and the item-list build use the wave's gather slots as scratch");
    static_assert(UPS * UNIT_BYTES + NROWS * WPITCH * 2 < 65536, "ds_read immediate offsets are 16 bits");
    static_assert(NB >= 2 && (NB - 2) * SLAB_PIECES <= 63 && (P - 1) * (NP + 1) <= 63 && P >= 2, "vmcnt is a 6-bit counter");
    static_assert(R >= 1 && R <= 4, "item encoding holds two bits of tile index");
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
};

// swizzle: f(row) such that the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots mod 256 B
template <int CH> __device__ __forceinline__ int gl_swz(int r)
{
    if constexpr (CH == 32) return (r & 3) | (((r >> 3) & 1) << 2);
    else return (r & 1) | (((r >> 3) & 1) << 1);
}

// ABL: compile-time ablations for tuning (only instantiated != 0 under -DAL3D_GLDS_ABLATE; see the dispatch)
template <int CIN, int COUT, int NW, int R, int UPS, int P, int NB, int ABL = 0>
__global__ __launch_bounds__(64 * (NW + 1)) void sp_conv_glds_kernel(const float* __restrict__ fin,
                                                                    const int* __restrict__ nbr, int pitch,
                                                                    const unsigned* __restrict__ tmask, int ntiles,
                                                                    const unsigned char* __restrict__ wpk,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    const float* __restrict__ residual, int relu,
                                                                    float* __restrict__ fout, int n_out, int io)
{
    using C = GldsCfg<CIN, COUT, NW, R, UPS, P, NB>;
    constexpr int KG = C::KG, CH = C::CH, UA = C::UA, NCC = C::NCC, TN = C::TN, NROWS = C::NROWS;
    constexpr int UNIT_BYTES = C::UNIT_BYTES, SPS = C::SPS, LPR = C::LPR, NP = C::NP, SLOT_BYTES = C::SLOT_BYTES;
    constexpr int T = C::T, LREG = C::LREG;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware placement (as sp_conv_wave2_kernel): each XCD gets one contiguous range of row tiles
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tile0 = wgt * T;
    const bool consumer = wave < NW;

    const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;
    const unsigned w_base = smem_base;
    const unsigned a_base = smem_base + C::W_BYTES + (consumer ? wave : 0) * C::A_WAVE_BYTES;
    const unsigned x_base = smem_base + C::W_BYTES + C::A_BYTES + (consumer ? wave : 0) * C::X_WAVE_BYTES;

    // ---- tap masks of the workgroup's tiles (one word each, from the rulebook): no scan, no barrier
    unsigned gmask = 0u, wm[R];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = tile0 + t;
        const unsigned m = tile < ntiles ? tmask[tile] : 0u;
        gmask |= m;
    }
    gmask = __builtin_amdgcn_readfirstlane(gmask);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int tile = tile0 + (consumer ? wave : 0) * R + r;
        wm[r] = __builtin_amdgcn_readfirstlane(tile < ntiles ? tmask[tile] : 0u);
    }
    const int nsteps = __builtin_popcount(gmask) * NCC;
    const int nslabs = (nsteps + SPS - 1) / SPS;

    if (!consumer) {
        // ================= producer wave: weight slabs -> a ring of NB LDS buffers, one barrier per slab.
        // Every slab is exactly SLAB_PIECES DMAs (units past the end re-fetch unit 0: harmless, never
        // read), so the counted wait retires slab s while slabs s+1 .. s+NB-2 stay in flight.
        unsigned rem = gmask;
        int g = 0;
        auto issue_slab = [&](int buf) {
            gl_static_for<UPS>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const int tap = rem ? __builtin_ctz(rem) : 0;
                const unsigned char* src = wpk + (size_t)(tap * KG + (rem ? g : 0)) * UNIT_BYTES + lane * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * C::SLAB_BYTES + q * UNIT_BYTES);
                gl_static_for<C::UNIT_PIECES>([&](auto PC) {
                    constexpr int pc = decltype(PC)::value;
                    __builtin_amdgcn_global_load_lds((gbl_void*)(src + pc * 1024), (lds_void*)(size_t)(dst + pc * 1024),
                                                     16, 0, 0);
                });
                if (rem && ++g == KG) { g = 0; rem &= rem - 1u; }
            });
        };
        int fill = 0;                                                       // ring position of the next slab to issue
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) {
            issue_slab(fill);
            fill = fill + 1 == NB ? 0 : fill + 1;
        }
        for (int s = 0; s < nslabs; ++s) {
            gl_wait_vm<(NB - 2) * C::SLAB_PIECES>();                         // slab s has landed
            __builtin_amdgcn_s_barrier();                                    // ... and slab s-1's buffer is free
            issue_slab(fill);
            fill = fill + 1 == NB ? 0 : fill + 1;
        }
        gl_wait_vm<0>();
        return;
    }

    // ================= consumer waves
    // ---- item list: (tap, chunk, tile) triples in processing order, compacted through the wave's scratch.
    // item = tap | chunk << 5 | tile << 7 | workgroup step << 9
    int lst[LREG];
    int nitems = 0;
    {
        unsigned short* scr16 = reinterpret_cast<unsigned short*>(smem + C::W_BYTES + wave * C::A_WAVE_BYTES);
#pragma unroll
        for (int q = 0; q < LREG; ++q) {
            const int c = q * 64 + lane;
            const int tap = c / (NCC * R), rm = c % (NCC * R), cc = rm / R, r = rm % R;
            unsigned mr = wm[0];
#pragma unroll
            for (int i = 1; i < R; ++i) mr = r == i ? wm[i] : mr;
            const bool valid = c < C::MAXITEMS && (gmask >> tap & 1u) && (mr >> tap & 1u);
            const int wgstep = __builtin_popcount(gmask & ((1u << tap) - 1u)) * NCC + cc;
            const unsigned long long bal = __ballot(valid);
            const int pos = nitems + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
            if (valid) scr16[pos] = (unsigned short)(tap | cc << 5 | r << 7 | wgstep << 9);
            nitems += __builtin_popcountll(bal);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < LREG; ++q) lst[q] = scr16[q * 64 + lane];
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): the list is in registers
        __builtin_amdgcn_wave_barrier();
    }
    auto get_item = [&](int i) -> int {                                      // wave-uniform i
        int v = lst[0];
#pragma unroll
        for (int q = 1; q < LREG; ++q) v = (i >> 6) == q ? lst[q] : v;
        return __builtin_amdgcn_readlane(v, i & 63);
    };

    f32x16 acc[R][TN];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;

    // gather side: this lane fetches chunk (s ^ f(row)) of rows NP*jg + i, i = 0..NP-1
    const int jg = lane / LPR, sg = lane % LPR;
    unsigned chk[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) chk[i] = (unsigned)((sg ^ gl_swz<CH>(NP * jg + i)) * 16);
    const unsigned x_lane = x_base + jg * (NP * 4);                          // its NP consecutive index entries
    // fragment side: lane (fr, fh) reads chunks 4*ua + 2*fh + h of row fr
    unsigned offA[UA][2];
#pragma unroll
    for (int ua = 0; ua < UA; ++ua)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            offA[ua][h] = (unsigned)((fr % NP) * 1024 + ((fr / NP) * LPR + ((4 * ua + 2 * fh + h) ^ gl_swz<CH>(fr))) * 16);
    const unsigned offB = w_base + (2 * fr + (fh ^ ((fr >> 3) & 1))) * 16;     // the image's swizzle (sp_pack_glds_kernel)
    const int wtile0 = tile0 + wave * R;

    // X(t): the 32 neighbour indices of item t -> index ring entry t & 1 (one DMA; items past the end read -1)
    auto issue_x = [&](int t) {
        const int* src = g_glds_neg1 + (lane & 31);
        if (t < nitems) {
            const int it = get_item(t);
            src = nbr + (int64_t)(it & 31) * pitch + (int64_t)(wtile0 + ((it >> 7) & 3)) * 32 + (lane & 31);
        }
        if constexpr (ABL != 4)
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(size_t)__builtin_amdgcn_readfirstlane(x_base + (t & 1) * 256),
                                             4, 0, 0);
    };
    // G(t): the NP row DMAs of item t -> gather slot t % P (slot passed in); needs X(t) landed
    const float* zrow = g_glds_zero;
    auto issue_g = [&](int t, int slot) {
        if constexpr (ABL == 4) return;
        int id[NP];
        if constexpr (NP == 4) {
            gl_i32x4 v;
            gl_lds_read_idx(v, x_lane + (t & 1) * 256);
            id[0] = v[0]; id[1] = v[1]; id[2] = v[2]; id[3] = v[3];
        } else {
            gl_i32x2 v;
            gl_lds_read_idx(v, x_lane + (t & 1) * 256);
            id[0] = v[0]; id[1] = v[1];
        }
        const int cc = t < nitems ? (get_item(t) >> 5) & 3 : 0;
        const unsigned sbase = a_base + slot * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if constexpr (ABL == 1) id[i] = -1;
            const char* rowp = id[i] >= 0 ? reinterpret_cast<const char*>(fin) + (int64_t)id[i] * (CIN * 4) + cc * (CH * 4)
                                          : reinterpret_cast<const char*>(zrow);
            __builtin_amdgcn_global_load_lds((gbl_void*)(rowp + chk[i]),
                                             (lds_void*)(size_t)__builtin_amdgcn_readfirstlane(sbase + i * 1024), 16, 0, 0);
        }
    };

    // stream of VMEM operations: X(0) | [X(1) G(0)] [X(2) G(1)] ...; the prologue issues the pairs 0 .. P-2
    issue_x(0);
#pragma unroll
    for (int t = 0; t < P - 1; ++t) {
        issue_x(t + 1);
        gl_wait_vm<1>();                                                     // X(t) (conservative: also G(t-1))
        issue_g(t, t);
    }

    f16x8 wh[UA][TN], wl[UA][TN];                                            // B fragments of the current (tap, chunk)
    int j = 0, slot = 0, pslot = P - 1, wbuf = 0, last_step = -1;
    for (int sl = 0; sl < nslabs; ++sl) {
        __builtin_amdgcn_s_barrier();
        const unsigned bslab = offB + wbuf * C::SLAB_BYTES;
        wbuf = wbuf + 1 == NB ? 0 : wbuf + 1;
        const int step_end = (sl + 1) * SPS;
        while (j < nitems) {
            const int it = get_item(j);
            const int wgstep = it >> 9;
            if (wgstep >= step_end) break;                                   // belongs to a later slab
            issue_x(j + P);
            if constexpr (ABL != 4) gl_wait_vm<NP + 1>();                    // X(j+P-1) has landed
            issue_g(j + P - 1, pslot);
            if constexpr (ABL != 4) gl_wait_vm<(P - 1) * (NP + 1)>();        // G(j) has landed
            const unsigned sA = a_base + slot * SLOT_BYTES;
            const unsigned bB = bslab + (wgstep - sl * SPS) * (UA * UNIT_BYTES);
            const bool newb = !C::BHOLD || wgstep != last_step;               // wave-uniform
            last_step = wgstep;
            const int rr = (it >> 7) & 3;
            if constexpr (ABL != 3) gl_static_for<R>([&](auto RR) {
                constexpr int r = decltype(RR)::value;
                if (rr != r) return;                                         // wave-uniform: one copy of the body per tile
                gl_static_for<UA>([&](auto U) {
                    constexpr int ua = decltype(U)::value;
                    gl_f32x4 lo, hi;
                    if (newb)
                        gl_lds_read_ab<TN, ua * UNIT_BYTES, NROWS * C::WPITCH>(lo, hi, wh[ua], wl[ua], sA + offA[ua][0],
                                                                               sA + offA[ua][1], bB);
                    else
                        gl_lds_read_a(lo, hi, sA + offA[ua][0], sA + offA[ua][1]);
                    f16x8 ah, al;
                    if (io & SP_IO_IN_PAIR) {                                  // pair rows: the fragment IS the operand pair
                        ah = __builtin_bit_cast(f16x8, lo);
                        al = __builtin_bit_cast(f16x8, hi);
                    } else {
                        gl_split8_f16(lo, hi, ah, al);
                    }
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        const f16x8 wd = gl_lift_down(wh[ua][jn]);
                        acc[r][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd, acc[r][jn], 0, 0, 0);     // smallest first
                        acc[r][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ua][jn], acc[r][jn], 0, 0, 0);
                        acc[r][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ua][jn], acc[r][jn], 0, 0, 0);
                    }
                });
            });
            slot = slot + 1 == P ? 0 : slot + 1;
            pslot = pslot + 1 == P ? 0 : pslot + 1;
            ++j;
        }
    }
    gl_wait_vm<0>();                                                         // the tail's dummy gathers

    // ---- epilogue (as sp_conv_wave2_kernel): transpose each 32 x 32 C tile through the wave's own LDS
    // (its gather slots: wave-private, no barrier) and move float4s over the contiguous 32-row block
    float* scr = reinterpret_cast<float*>(smem + C::W_BYTES + wave * C::A_WAVE_BYTES);
    constexpr int EP_PITCH = C::EP_PITCH;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int wrow0 = (wtile0 + r) * 32;
        if (wrow0 >= n_out) break;                                           // wave-uniform
#pragma unroll
        for (int j2 = 0; j2 < TN; ++j2) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                scr[((e & 3) + 8 * (e >> 2) + 4 * fh) * EP_PITCH + fr] = acc[r][j2][e];
            __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            sp_store_tile<COUT, EP_PITCH>(scr, lane, j2, wrow0, n_out, scale, shift, residual, relu, fout, io);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- weight image: [K][KG][2 planes][NROWS][2 chunks of 8 f16]; chunk c of row n sits at position
// c ^ ((n >> 3) & 1) (bank swizzle instead of padding: conflict-free for the b128 lane groups); rows >= Cout zero
__global__ void sp_pack_glds_kernel(const unsigned short* __restrict__ planes, int cout, int K, int cin, int nrows,
                                    unsigned short* __restrict__ out, int64_t total)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t % 8);
    const int pos = (int)((t / 8) % 2);
    int64_t r = t / 16;
    const int n = (int)(r % nrows); r /= nrows;
    const int pl = (int)(r % 2); r /= 2;
    const int kg = cin / 16;
    const int g = (int)(r % kg);
    const int tap = (int)(r / kg);
    const int c = pos ^ ((n >> 3) & 1);
    unsigned short v = 0;
    if (n < cout) v = planes[(((int64_t)pl * cout + n) * K + tap) * cin + 16 * g + 8 * c + e];
    out[t] = v;
}

extern "C" int64_t al3d_sp_pack_glds_f16x3_elems(int cout, int K, int cin)
{
    if (cout < 1 || K < 1 || K > 27 || cin < 16 || cin % 16) return -1;
    const int nrows = (cout + 31) / 32 * 32;
    return (int64_t)K * (cin / 16) * 2 * nrows * 16;
}

extern "C" int al3d_sp_pack_glds_f16x3(const void* planes_f16x2, int cout, int K, int cin, void* out_image, void* stream)
{
    const int64_t total = al3d_sp_pack_glds_f16x3_elems(cout, K, cin);
    AL3D_REQUIRE(total > 0, "al3d_sp_pack_glds_f16x3: unsupported shape Cout=%d K=%d Cin=%d", cout, K, cin);
    AL3D_REQUIRE(planes_f16x2 && out_image, "al3d_sp_pack_glds_f16x3: null pointer");
    const int nrows = (cout + 31) / 32 * 32;
    hipLaunchKernelGGL(sp_pack_glds_kernel, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)planes_f16x2, cout, K, cin, nrows, (unsigned short*)out_image, total);
    AL3D_CHECK_LAUNCH("sp_pack_glds_kernel");
    return AL3D_OK;
}

#define GLDS_DISPATCH_ABL(CI, CO, NW, R, UPS, P, NB, ABL)                                              \
    if (cin == CI && cout == CO) {                                                                      \
        hipLaunchKernelGGL((sp_conv_glds_kernel<CI, CO, NW, R, UPS, P, NB, ABL>),                       \
                           dim3((unsigned)al3d_cdiv(ntiles, NW * R)), dim3(64 * (NW + 1)), 0, s, fin, nbr, nbr_pitch, \
                           tile_mask, ntiles, (const unsigned char*)wgt_image, scale, shift, residual, relu, fout, n_out, io); \
        AL3D_CHECK_LAUNCH("sp_conv_glds_kernel");                                                       \
        return AL3D_OK;                                                                                 \
    }
#define GLDS_DISPATCH(CI, CO, NW, R, UPS, P, NB) GLDS_DISPATCH_ABL(CI, CO, NW, R, UPS, P, NB, 0)

static int sp_conv_glds_impl(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                             const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                             const float* residual, int relu, float* fout, int n_out, int io, void* stream);

extern "C" int al3d_sp_conv_glds_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                                       const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                                       const float* residual, int relu, float* fout, int n_out, void* stream)
{
    return sp_conv_glds_impl(fin, nbr, nbr_pitch, tile_mask, K, wgt_image, cin, cout, scale, shift, residual, relu, fout, n_out,
                             0, stream);
}

// ... with the row formats of sp_rows.h: io bit 0 = input pair rows, bit 1 = write pair rows, bit 2 = residual pair rows
extern "C" int al3d_sp_conv_glds_f16x3_io(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                                          const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                                          const float* residual, int relu, float* fout, int n_out, int io, void* stream)
{
    AL3D_REQUIRE(io >= 0 && io < 8 && cout % 8 == 0, "al3d_sp_conv_glds_f16x3_io: bad io flags / channels");
    return sp_conv_glds_impl(fin, nbr, nbr_pitch, tile_mask, K, wgt_image, cin, cout, scale, shift, residual, relu, fout, n_out,
                             io, stream);
}

static int sp_conv_glds_impl(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask, int K,
                             const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                             const float* residual, int relu, float* fout, int n_out, int io, void* stream)
{
    AL3D_REQUIRE(K >= 1 && K <= 27 && n_out >= 0, "al3d_sp_conv_glds_f16x3: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && tile_mask && wgt_image && fout, "al3d_sp_conv_glds_f16x3: null pointer");
    AL3D_REQUIRE(scale, "al3d_sp_conv_glds_f16x3: scale carries the weight exponent and is required");
    AL3D_REQUIRE(nbr_pitch >= n_out && nbr_pitch % 256 == 0, "al3d_sp_conv_glds_f16x3: nbr_pitch must be al3d_sp_table_pitch(n_out)");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)al3d_cdiv(n_out, 32);
#ifdef AL3D_GLDS_ABLATE
    // tuning build only (make EXTRA=-DAL3D_GLDS_ABLATE): AL3D_GLDS_ABL / AL3D_GLDS_CFG pick an ablation / a shape
    {
        const char* e = getenv("AL3D_GLDS_ABL");
        const char* c = getenv("AL3D_GLDS_CFG");
        const int abl = e ? atoi(e) : 0, cfgi = c ? atoi(c) : 0;
#define GLDS_ABL_SET(A) { GLDS_DISPATCH_ABL(32, 32, 4, 4, 4, 3, 2, A) GLDS_DISPATCH_ABL(64, 64, 4, 2, 2, 3, 2, A) GLDS_DISPATCH_ABL(128, 128, 4, 1, 2, 2, 2, A) GLDS_DISPATCH_ABL(16, 16, 4, 4, 4, 4, 2, A) }
        if (abl == 1) GLDS_ABL_SET(1)
        if (abl == 3) GLDS_ABL_SET(3)
        if (abl == 4) GLDS_ABL_SET(4)
        if (cfgi == 1) { GLDS_DISPATCH(32, 32, 7, 1, 4, 2, 2) GLDS_DISPATCH(64, 64, 7, 1, 2, 2, 2) GLDS_DISPATCH(128, 128, 5, 1, 2, 2, 2) GLDS_DISPATCH(16, 16, 8, 1, 4, 3, 2) }
        if (cfgi == 2) { GLDS_DISPATCH(32, 32, 6, 2, 4, 2, 2) GLDS_DISPATCH(64, 64, 6, 2, 2, 2, 2) GLDS_DISPATCH(128, 128, 6, 1, 2, 2, 2) GLDS_DISPATCH(16, 16, 8, 2, 4, 3, 2) }
        if (cfgi == 3) { GLDS_DISPATCH(32, 32, 4, 2, 4, 3, 2) GLDS_DISPATCH(64, 64, 4, 2, 2, 3, 2) GLDS_DISPATCH(128, 128, 4, 1, 2, 3, 2) GLDS_DISPATCH(16, 16, 6, 2, 4, 4, 2) }
        if (cfgi == 4) { GLDS_DISPATCH(32, 32, 5, 1, 4, 3, 2) GLDS_DISPATCH(64, 64, 5, 1, 2, 3, 2) GLDS_DISPATCH(128, 128, 3, 1, 2, 3, 2) GLDS_DISPATCH(16, 16, 6, 1, 4, 4, 2) }
    }
#endif
    // LDS <= 80 KB per workgroup: two workgroups share a CU (one's prologue / epilogue / barrier waits overlap
    // the other's main loop)
    // measured on the real rulebooks (tools/glds_ablate.sh): one tile per wave and as many waves as the LDS allows
    // beat several tiles per wave (fewer waves) everywhere
    GLDS_DISPATCH(16, 16, 8, 1, 4, 3, 2) GLDS_DISPATCH(16, 32, 8, 1, 4, 3, 2) GLDS_DISPATCH(32, 32, 7, 1, 4, 2, 2) GLDS_DISPATCH(32, 64, 7, 1, 2, 2, 2)
    GLDS_DISPATCH(64, 64, 7, 1, 2, 2, 2) GLDS_DISPATCH(64, 128, 5, 1, 2, 2, 2) GLDS_DISPATCH(128, 128, 5, 1, 2, 2, 2)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_glds_f16x3: unsupported channel pair %d -> %d", cin, cout);
}
