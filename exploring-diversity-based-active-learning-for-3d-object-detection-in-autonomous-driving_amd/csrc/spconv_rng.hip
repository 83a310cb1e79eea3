// Submanifold sparse convolution, f16x3 arithmetic, RANGE gather by LDS-DMA (gfx950).
//
// spconv_glds.hip fetches, per (32-row output tile, tap), the 32 neighbour rows of the tap -- 27 x 32 rows per
// tile, most of them fetched three times: the three kx taps of one (kz, ky) reference almost the same input
// rows, shifted by one site.  From level 1 on the rows of a level are in raster order (z, y, x), so the
// neighbours of 32 consecutive output rows under the three kx taps of a (kz, ky) GROUP lie in one short
// contiguous index range [lo, lo + L): measured on the real rulebooks (tools/probe_kx_ranges.py) L has median
// 33-34, p90 37-39 and is <= 48 for 96-97 % of the live (tile, group) pairs, against 3 x 32 = 96 rows gathered per
// group by the per-tap scheme.  Here a group item stages rows lo .. lo + CAP - 1 ONCE -- CAP / 8 fully coalesced
// 1 KiB DMA pieces, no per-row address, no dependence on the neighbour indices -- and the three taps read their A
// fragments from it at LDS row (index - lo); rows without a neighbour read a shared zero row.  The (lo, L) table
// comes with the rulebook (al3d_sp_tile_ranges, one pass over the level's table, shared by its SubM layers).
// Groups with L > CAP (tiles straddling a z plane or a long gap; ~7 % at the shipped CAP = 40) fall back, inside
// the same item, to a per-row gather of each tap into the same slot (drains the DMA queue: correct, slower).
//
// Everything else is the LDS-DMA kernel's: a producer wave streams weight slabs (one slab = the 3 taps x 2
// sixteen-channel units of a group item) through a ring of two LDS buffers, one raw s_barrier per slab; consumer
// waves own one tile each, keep P slots of CAP rows and run P-1 items ahead; all main-loop LDS reads are asm
// blocks fused with their waits; counted s_waitcnt vmcnt.  Swizzle: chunk c of staged row r sits at position
// c ^ ((r >> 1) & 7) (applied on the source side of the DMA), conflict-free for the b128 lane groups whenever
// the 32 rows of a fragment are consecutive.
//
// Accumulation order per output row: group, channel chunk, kx, unit -- for Cin = 32 (one chunk) exactly the tap,
// unit order of sp_conv_wave2 / sp_conv_glds: bit-identical.
#include "glds_common.h"
#include "sp_rows.h"

template <int CIN, int COUT, int NW, int P, int CAP>
struct RngCfg {
    static constexpr int KG = CIN / 16;
    static constexpr int UA = 2;                          // 16-channel units per 32-channel chunk
    static constexpr int NCC = CIN / 32;
    static constexpr int TN = COUT / 32;
    static constexpr int NROWS = TN * 32;
    static constexpr int UNIT_BYTES = 2 * NROWS * 32;     // two planes, 32-byte rows (the glds weight image)
    static constexpr int UNIT_PIECES = UNIT_BYTES / 1024;
    static constexpr int SLAB_UNITS = 3 * UA;
    static constexpr int SLAB_BYTES = SLAB_UNITS * UNIT_BYTES;
    static constexpr int SLAB_PIECES = SLAB_UNITS * UNIT_PIECES;
    static constexpr int W_BYTES = 2 * SLAB_BYTES;
    static constexpr int NPC = CAP / 8;                   // DMA pieces per range (8 rows x 128 B each)
    static constexpr int SLOT_BYTES = CAP * 128;
    static constexpr int A_WAVE_BYTES = P * SLOT_BYTES;
    static constexpr int A_BYTES = NW * A_WAVE_BYTES;
    static constexpr int X_ENTRY = 512;                   // tap 0 | tap 1 | tap 2 | (duplicate) x 32 indices
    static constexpr int X_WAVE_BYTES = 2 * X_ENTRY;
    static constexpr int ZERO_OFF = W_BYTES + A_BYTES + NW * X_WAVE_BYTES;
    static constexpr int SMEM_BYTES = ZERO_OFF + 128;
    static constexpr int NV = 2 + NPC;                    // VMEM operations per item
    static constexpr int WAITN = (P - 1) * NV < NPC + NV ? (P - 1) * NV : NPC + NV;
    static constexpr int EP_PITCH = 40;
    static_assert(CIN % 32 == 0 && COUT % 32 == 0 && CAP % 8 == 0 && CAP >= 32, "shape");
    static_assert(P >= 2 && WAITN <= 63 && SLAB_PIECES <= 63, "vmcnt is a 6-bit counter");
    static_assert(A_WAVE_BYTES >= 32 * EP_PITCH * 4, "the epilogue transposes through the wave's slots");
    static_assert(W_BYTES + SLAB_BYTES < 65536 + 32768, "ds_read immediate offsets");
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void rg_lds_read_idx3(int& i0, int& i1, int& i2, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:128\n\tds_read_b32 %2, %3 offset:256\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(i0), "=&v"(i1), "=&v"(i2) : "v"(addr) : "memory");
}
__device__ __forceinline__ void rg_lds_read_idx1(int& i0, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(i0) : "v"(addr) : "memory");
}

// One pipelined unit: the ds_reads of the NEXT unit's A / B fragments interleaved with the MFMAs of the current
// unit, closed by the lgkmcnt wait -- one asm block, so no register is visible to hipcc while it is in flight.
// (`s_nop 1`: the operands come from VALU instructions right before the block -- 2 wait states to an MFMA read.)
template <int TN, int OFF, int PL>
__device__ __forceinline__ void rg_read_next_mfma(gl_f32x4& nlo, gl_f32x4& nhi, f16x8 (&nwh)[TN], f16x8 (&nwl)[TN],
                                                  f32x16 (&acc)[TN], const f16x8& al, const f16x8& ah,
                                                  const f16x8 (&wd)[TN], const f16x8 (&wl)[TN], const f16x8 (&wh)[TN],
                                                  unsigned a0, unsigned a1, unsigned b)
{
    static_assert(TN == 1 || TN == 2, "tile counts of the built channel pairs");
    if constexpr (TN == 1)
        asm volatile("s_nop 1\n\t"
                     "ds_read_b128 %0, %10\n\tds_read_b128 %1, %11\n\t"
                     "v_mfma_f32_32x32x16_f16 %4, %5, %7, %4\n\t"
                     "ds_read_b128 %2, %12 offset:%13\n\tds_read_b128 %3, %12 offset:%14\n\t"
                     "v_mfma_f32_32x32x16_f16 %4, %6, %8, %4\n\t"
                     "v_mfma_f32_32x32x16_f16 %4, %6, %9, %4\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(nlo), "=&v"(nhi), "=&v"(nwh[0]), "=&v"(nwl[0]), "+v"(acc[0])
                     : "v"(al), "v"(ah), "v"(wd[0]), "v"(wl[0]), "v"(wh[0]), "v"(a0), "v"(a1), "v"(b), "n"(OFF), "n"(OFF + PL)
                     : "memory");
    else
        asm volatile("s_nop 1\n\t"
                     "ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\t"
                     "v_mfma_f32_32x32x16_f16 %6, %8, %10, %6\n\t"
                     "v_mfma_f32_32x32x16_f16 %7, %8, %13, %7\n\t"
                     "ds_read_b128 %2, %18 offset:%19\n\tds_read_b128 %3, %18 offset:%20\n\t"
                     "v_mfma_f32_32x32x16_f16 %6, %9, %11, %6\n\t"
                     "v_mfma_f32_32x32x16_f16 %7, %9, %14, %7\n\t"
                     "ds_read_b128 %4, %18 offset:%21\n\tds_read_b128 %5, %18 offset:%22\n\t"
                     "v_mfma_f32_32x32x16_f16 %6, %9, %12, %6\n\t"
                     "v_mfma_f32_32x32x16_f16 %7, %9, %15, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(nlo), "=&v"(nhi), "=&v"(nwh[0]), "=&v"(nwl[0]), "=&v"(nwh[1]), "=&v"(nwl[1]), "+v"(acc[0]), "+v"(acc[1])
                     : "v"(al), "v"(ah), "v"(wd[0]), "v"(wl[0]), "v"(wh[0]), "v"(wd[1]), "v"(wl[1]), "v"(wh[1]),
                       "v"(a0), "v"(a1), "v"(b), "n"(OFF), "n"(OFF + PL), "n"(OFF + 1024), "n"(OFF + PL + 1024)
                     : "memory");
}

template <int CIN, int COUT, int NW, int P, int CAP>
__global__ __launch_bounds__(64 * (NW + 1)) void sp_conv_rng_kernel(const float* __restrict__ fin,
                                                                   const int* __restrict__ nbr, int pitch,
                                                                   const unsigned* __restrict__ tmask,
                                                                   const int2* __restrict__ rng, int ntiles,
                                                                   const unsigned char* __restrict__ wpk,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ residual, int relu,
                                                                   float* __restrict__ fout, int n_out, int io)
{
    using C = RngCfg<CIN, COUT, NW, P, CAP>;
    constexpr int KG = C::KG, UA = C::UA, NCC = C::NCC, TN = C::TN, NROWS = C::NROWS, UNIT_BYTES = C::UNIT_BYTES;
    constexpr int NPC = C::NPC, SLOT_BYTES = C::SLOT_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware placement (as sp_conv_glds_kernel): each XCD gets one contiguous range of row tiles
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tile0 = wgt * NW;
    const bool consumer = wave < NW;

    const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;
    const unsigned a_base = smem_base + C::W_BYTES + (consumer ? wave : 0) * C::A_WAVE_BYTES;
    const unsigned x_base = smem_base + C::W_BYTES + C::A_BYTES + (consumer ? wave : 0) * C::X_WAVE_BYTES;
    const unsigned zero_base = smem_base + C::ZERO_OFF;

    // the shared zero row (read by fragment lanes without a neighbour)
    if (tid < 32) reinterpret_cast<float*>(smem + C::ZERO_OFF)[tid] = 0.f;
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // ---- live (kz, ky) groups of the workgroup's tiles and of this wave's tile, from the rulebook's tap masks
    auto groups_of = [](unsigned m27) -> unsigned {
        unsigned g = 0u;
#pragma unroll
        for (int i = 0; i < 9; ++i) g |= ((m27 >> (3 * i)) & 7u) ? 1u << i : 0u;
        return g;
    };
    unsigned gm27 = 0u;
#pragma unroll
    for (int t = 0; t < NW; ++t) gm27 |= tile0 + t < ntiles ? tmask[tile0 + t] : 0u;
    const unsigned gmask = __builtin_amdgcn_readfirstlane(groups_of(gm27));
    const int nsteps = __builtin_popcount(gmask) * NCC;

    if (!consumer) {
        // ================= producer wave: one slab per workgroup step (group, chunk) = 3 taps x UA units
        unsigned rem = gmask;
        int cc = 0;
        auto issue_slab = [&](int buf) {
            const int g = rem ? __builtin_ctz(rem) : 0;
            gl_static_for<3>([&](auto KX) {
                constexpr int kx = decltype(KX)::value;
                const unsigned char* src = wpk + (size_t)((3 * g + kx) * KG + (rem ? cc : 0) * UA) * UNIT_BYTES + lane * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * C::SLAB_BYTES + kx * UA * UNIT_BYTES);
                gl_static_for<UA * C::UNIT_PIECES>([&](auto PC) {
                    constexpr int pc = decltype(PC)::value;
                    __builtin_amdgcn_global_load_lds((gbl_void*)(src + pc * 1024), (lds_void*)(size_t)(dst + pc * 1024),
                                                     16, 0, 0);
                });
            });
            if (rem && ++cc == NCC) { cc = 0; rem &= rem - 1u; }
        };
        issue_slab(0);
        for (int s = 0; s < nsteps; ++s) {
            gl_wait_vm<0>();                                                 // slab s has landed
            __builtin_amdgcn_s_barrier();                                    // ... and slab s-1's buffer is free
            issue_slab((s + 1) & 1);
        }
        gl_wait_vm<0>();
        return;
    }

    // ================= consumer waves: one tile each
    const int tile = tile0 + wave;
    const bool tile_ok = tile < ntiles;
    const unsigned wmask = __builtin_amdgcn_readfirstlane(tile_ok ? groups_of(tmask[tile]) : 0u);
    int my_lo = 0, my_len = 0;                                               // lane g holds the range of group g
    if (lane < 9 && tile_ok) {
        const int2 r = rng[(int64_t)tile * 9 + lane];
        my_lo = r.x; my_len = r.y;
    }

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // DMA side of a range: lane (j, s) of piece i fetches chunk s ^ f(r) of staged row r = 8 i + j
    const int jd = lane >> 3, sd = lane & 7;
    const int chk_even = (sd ^ (jd >> 1)) * 16;                              // f(8 i + j) = (4 i + (j >> 1)) & 7
    const int chk_odd = (sd ^ (jd >> 1) ^ 4) * 16;
    const unsigned offB = smem_base + (2 * fr + (fh ^ ((fr >> 3) & 1))) * 16;  // the weight image's swizzle

    // three cursors over the wave's items (group ascending, chunk inner): index requests, range requests, consumption
    unsigned xrem = wmask, irem = wmask, crem = wmask;
    int xcc = 0, icc = 0, ccc = 0;

    // X(t): the 3 x 32 neighbour indices of item t's group -> index ring entry t & 1 (two DMAs; absent items read -1)
    auto issue_x = [&](int entry) {
        const int* s0 = g_glds_neg1 + (lane & 31);
        const int* s1 = s0;
        if (xrem) {
            const int g = __builtin_ctz(xrem);
            s0 = nbr + (int64_t)(3 * g + fh) * pitch + (int64_t)tile * 32 + (lane & 31);
            s1 = nbr + (int64_t)(3 * g + 2) * pitch + (int64_t)tile * 32 + (lane & 31);
        }
        const unsigned d = __builtin_amdgcn_readfirstlane(x_base + entry * C::X_ENTRY);
        __builtin_amdgcn_global_load_lds((gbl_void*)s0, (lds_void*)(size_t)d, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)s1, (lds_void*)(size_t)(d + 256), 4, 0, 0);
        if (xrem && ++xcc == NCC) { xcc = 0; xrem &= xrem - 1u; }
    };
    // G(t): rows lo .. lo + CAP - 1 (clamped to the range) of chunk cc -> slot t % P; irregular / absent items: zero
    // row (the count of NPC requests per item is what the counted waits rely on).  A clamped request stages row
    // `last` again at a staged row >= len: never read (fragment lanes only address rows < len).
    auto issue_g = [&](int slot) {
        int lo = 0, len = 0;
        if (irem) {
            const int g = __builtin_ctz(irem);
            lo = __builtin_amdgcn_readlane(my_lo, g);
            len = __builtin_amdgcn_readlane(my_len, g);
        }
        const bool regular = len > 0 && len <= CAP;                          // wave-uniform
        const char* base = regular ? reinterpret_cast<const char*>(fin) + (int64_t)lo * (CIN * 4) + icc * 128
                                   : reinterpret_cast<const char*>(g_glds_zero);
        const int stride = regular ? CIN * 4 : 0;
        const int last = regular ? len - 1 : 0;
        const unsigned sbase = a_base + slot * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            int r = 8 * i + jd;
            r = r < last ? r : last;
            const unsigned off = (unsigned)(r * stride + ((i & 1) ? chk_odd : chk_even));
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)(size_t)__builtin_amdgcn_readfirstlane(sbase + i * 1024),
                                             16, 0, 0);
        }
        if (irem && ++icc == NCC) { icc = 0; irem &= irem - 1u; }
    };

    // Stream of VMEM operations.  The end of iteration t issues X(t+2) into the index entry and G(t+P) into the slot
    // that item t just released; the prologue is the same pattern for the virtual iterations -P .. -1:
    //     G(0) | X(0) G(1) | X(1) G(2) | ... | X(P-2) G(P-1)   ||   X(2)... wait for item t: everything up to X(t), G(t)
    // After G(t) follow (P-1) NV operations, after X(t) follow NPC + NV: waiting for vmcnt <= WAITN = the smaller of
    // the two retires both (operations retire in order).
    static_assert(P >= 2, "the index ring has two entries: X runs two items ahead, G at least as far");
#pragma unroll
    for (int u = -P; u < 0; ++u) {
        if (u + 2 >= 0 && u + 2 < 2) issue_x(u + 2);
        issue_g(u + P);
    }

    int slot = 0, xe = 0;
    for (int sl = 0; sl < nsteps; ++sl) {
        __builtin_amdgcn_s_barrier();
        if (!crem) continue;                                                 // this wave's items are done (uniform)
        const int g = __builtin_ctz(crem);
        const int wgstep = __builtin_popcount(gmask & ((1u << g) - 1u)) * NCC + ccc;
        if (wgstep != sl) continue;                                          // a group of other tiles only
        const unsigned bslab = offB + (sl & 1) * C::SLAB_BYTES;
        const int lo = __builtin_amdgcn_readlane(my_lo, g);
        const int len = __builtin_amdgcn_readlane(my_len, g);
        const unsigned sA = a_base + slot * SLOT_BYTES;
        const unsigned xa = x_base + xe * C::X_ENTRY + fr * 4;
#ifdef RNG_ABL_NOIRR
        if (true) {
#else
        if (len <= CAP) {
#endif
            gl_wait_vm<C::WAITN>();                                          // X(t), G(t) have landed (see the order above)
            int id[3];
            rg_lds_read_idx3(id[0], id[1], id[2], xa);
            // fragment addresses of the six units (kx, ua) of the item
            unsigned ua0[3 * UA], ua1[3 * UA];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool ok = id[kx] >= 0;
                const int local = id[kx] - lo;
                const unsigned rowa = ok ? sA + (unsigned)local * 128u : zero_base;
                const unsigned x = ok ? (unsigned)((2 * fh) ^ ((local >> 1) & 7)) : 0u;
#pragma unroll
                for (int ua = 0; ua < UA; ++ua) {
                    ua0[kx * UA + ua] = rowa + ((x ^ (4 * ua)) << 4);
                    ua1[kx * UA + ua] = rowa + ((x ^ (4 * ua + 1)) << 4);
                }
            }
            // software pipeline over the units: unit u+1's fragments are read while unit u's MFMAs run
            gl_f32x4 vlo[2], vhi[2];
            f16x8 wh[2][TN], wl[2][TN];
            gl_lds_read_ab<TN, 0, NROWS * 32>(vlo[0], vhi[0], wh[0], wl[0], ua0[0], ua1[0], bslab);
            gl_static_for<3 * UA>([&](auto U) {
                constexpr int u = decltype(U)::value;
                constexpr int cur = u & 1, nxt = cur ^ 1;
                f16x8 ah, al, wd[TN];
                if (io & SP_IO_IN_PAIR) {                                      // pair rows: the fragment IS the operand pair
                    ah = __builtin_bit_cast(f16x8, vlo[cur]);
                    al = __builtin_bit_cast(f16x8, vhi[cur]);
                } else {
                    gl_split8_f16(vlo[cur], vhi[cur], ah, al);
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) wd[jn] = gl_lift_down(wh[cur][jn]);
                if constexpr (u + 1 < 3 * UA) {
                    rg_read_next_mfma<TN, (u + 1) * UNIT_BYTES, NROWS * 32>(vlo[nxt], vhi[nxt], wh[nxt], wl[nxt], acc, al, ah, wd,
                                                                           wl[cur], wh[cur], ua0[u + 1], ua1[u + 1], bslab);
                } else {
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd[jn], acc[jn], 0, 0, 0);     // smallest first
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[cur][jn], acc[jn], 0, 0, 0);
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[cur][jn], acc[jn], 0, 0, 0);
                    }
                }
            });
        } else {
            // irregular group: per-row gather of each tap into rows 0..31 of the slot (drains the queue)
            gl_static_for<3>([&](auto KX) {
                constexpr int kx = decltype(KX)::value;
                gl_wait_vm<0>();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int idr;
                    rg_lds_read_idx1(idr, x_base + xe * C::X_ENTRY + kx * 128 + (8 * i + jd) * 4);
                    const char* src = idr >= 0 ? reinterpret_cast<const char*>(fin) + (int64_t)idr * (CIN * 4) + ccc * 128 +
                                                     ((i & 1) ? chk_odd : chk_even)
                                               : reinterpret_cast<const char*>(g_glds_zero) + sd * 16;
                    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(size_t)__builtin_amdgcn_readfirstlane(sA + i * 1024),
                                                     16, 0, 0);
                }
                gl_wait_vm<0>();
                int idf;
                rg_lds_read_idx1(idf, xa + kx * 128);
                const bool ok = idf >= 0;
                const unsigned rowa = ok ? sA + (unsigned)fr * 128u : zero_base;
                const unsigned x = ok ? (unsigned)((2 * fh) ^ ((fr >> 1) & 7)) : 0u;
                gl_static_for<UA>([&](auto U) {
                    constexpr int ua = decltype(U)::value;
                    gl_f32x4 vlo, vhi;
                    f16x8 wh[TN], wl[TN];
                    gl_lds_read_ab<TN, (kx * UA + ua) * UNIT_BYTES, NROWS * 32>(vlo, vhi, wh, wl, rowa + ((x ^ (4 * ua)) << 4),
                                                                                rowa + ((x ^ (4 * ua + 1)) << 4), bslab);
                    f16x8 ah, al;
                    if (io & SP_IO_IN_PAIR) {
                        ah = __builtin_bit_cast(f16x8, vlo);
                        al = __builtin_bit_cast(f16x8, vhi);
                    } else {
                        gl_split8_f16(vlo, vhi, ah, al);
                    }
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        const f16x8 wd = gl_lift_down(wh[jn]);
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd, acc[jn], 0, 0, 0);
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[jn], acc[jn], 0, 0, 0);
                        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[jn], acc[jn], 0, 0, 0);
                    }
                });
            });
        }
        issue_x(xe);                                                         // X(t+2) -> the entry item t just released
        issue_g(slot);                                                       // G(t+P) -> the slot item t just released
        if (++ccc == NCC) { ccc = 0; crem &= crem - 1u; }
        slot = slot + 1 == P ? 0 : slot + 1;
        xe ^= 1;
    }
    gl_wait_vm<0>();                                                         // the tail's dummy requests

    // ---- epilogue (as sp_conv_glds_kernel): transpose each 32 x 32 C tile through the wave's own LDS
    float* scr = reinterpret_cast<float*>(smem + C::W_BYTES + wave * C::A_WAVE_BYTES);
    constexpr int EP_PITCH = C::EP_PITCH;
    const int wrow0 = tile * 32;
    if (wrow0 >= n_out) return;                                              // wave-uniform
#pragma unroll
    for (int j2 = 0; j2 < TN; ++j2) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            scr[((e & 3) + 8 * (e >> 2) + 4 * fh) * EP_PITCH + fr] = acc[j2][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        sp_store_tile<COUT, EP_PITCH>(scr, lane, j2, wrow0, n_out, scale, shift, residual, relu, fout, io);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- (lo, len) of every (32-row tile, (kz, ky) group): the contiguous index range covering the valid neighbours
// of the group's three kx taps; len = 0: no valid neighbour.  One half-wave per tile, 27 coalesced 128-byte reads.
__global__ __launch_bounds__(256) void sp_tile_ranges_kernel(const int* __restrict__ nbr, int64_t pitch, int n_out,
                                                             int ntiles, int2* __restrict__ rng)
{
    const int tile = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r = threadIdx.x & 31;
    if (tile >= ntiles) return;                                              // half-wave uniform
    const int row = tile * 32 + r;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        int mn = 0x7fffffff, mx = -1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int v = row < n_out ? nbr[(int64_t)(3 * g + kx) * pitch + row] : -1;
            if (v >= 0) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            const int omn = __shfl_xor(mn, d, 32), omx = __shfl_xor(mx, d, 32);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if (r == 0) rng[(int64_t)tile * 9 + g] = mx >= 0 ? make_int2(mn, mx - mn + 1) : make_int2(0, 0);
    }
}

extern "C" int al3d_sp_tile_ranges(const int* nbr, int64_t nbr_pitch, int K, int n_out, int* out_rng, void* stream)
{
    AL3D_REQUIRE(K == 27 && n_out >= 0 && nbr_pitch >= n_out, "al3d_sp_tile_ranges: 27-tap tables only");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(nbr && out_rng, "al3d_sp_tile_ranges: null pointer");
    const int ntiles = (int)al3d_cdiv(n_out, 32);
    hipLaunchKernelGGL(sp_tile_ranges_kernel, dim3((unsigned)al3d_cdiv(ntiles, 8)), dim3(256), 0, (hipStream_t)stream, nbr,
                       nbr_pitch, n_out, ntiles, (int2*)out_rng);
    AL3D_CHECK_LAUNCH("sp_tile_ranges_kernel");
    return AL3D_OK;
}

#define RNG_DISPATCH(CI, CO, NW, P, CAP)                                                                      \
    if (cin == CI && cout == CO) {                                                                              \
        hipLaunchKernelGGL((sp_conv_rng_kernel<CI, CO, NW, P, CAP>), dim3((unsigned)al3d_cdiv(ntiles, NW)),     \
                           dim3(64 * (NW + 1)), 0, s, fin, nbr, nbr_pitch, tile_mask, (const int2*)tile_rng, ntiles, \
                           (const unsigned char*)wgt_image, scale, shift, residual, relu, fout, n_out, io);     \
        AL3D_CHECK_LAUNCH("sp_conv_rng_kernel");                                                                \
        return AL3D_OK;                                                                                         \
    }

// Same contract as al3d_sp_conv_glds_f16x3 (tiled 27-tap SubM table, weight image of al3d_sp_pack_glds_f16x3) plus
// tile_rng from al3d_sp_tile_ranges on the same table.  Rows of the level must be in raster order for the ranges to
// be short (any order is CORRECT: long ranges take the per-row path).
extern "C" int al3d_sp_conv_rng_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                                      const int* tile_rng, int K, const void* wgt_image, int cin, int cout,
                                      const float* scale, const float* shift, const float* residual, int relu,
                                      float* fout, int n_out, int io, void* stream)
{
    AL3D_REQUIRE(K == 27 && n_out >= 0, "al3d_sp_conv_rng_f16x3: 27-tap submanifold layers only");
    AL3D_REQUIRE(io >= 0 && io < 8, "al3d_sp_conv_rng_f16x3: bad io flags");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && tile_mask && tile_rng && wgt_image && fout, "al3d_sp_conv_rng_f16x3: null pointer");
    AL3D_REQUIRE(scale, "al3d_sp_conv_rng_f16x3: scale carries the weight exponent and is required");
    AL3D_REQUIRE(nbr_pitch >= n_out && nbr_pitch % 256 == 0, "al3d_sp_conv_rng_f16x3: nbr_pitch must be al3d_sp_table_pitch(n_out)");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)al3d_cdiv(n_out, 32);
    // shapes: as many consumer waves as the LDS holds with two slots each -- the kernel is bound by the latency chain
    // of a wave's item times the resident waves (7 -> 10 -> 12 waves: 968 -> 852 -> 795 us on the 32 -> 32 layers),
    // so the range cap is 40 rows (5 KB slots; ~7 % of the groups take the per-row path) rather than 48
    RNG_DISPATCH(32, 32, 12, 2, 40)
    RNG_DISPATCH(64, 64, 10, 2, 40)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_rng_f16x3: no kernel for Cin=%d Cout=%d", cin, cout);
}

// ---- row-format conversions (sp_rows.h): [n][C] f32 <-> pair rows, one thread per (row, 8-channel group)
__global__ __launch_bounds__(256) void sp_rows_convert_kernel(const float* __restrict__ in, int64_t groups, int to_pair,
                                                              float* __restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const float* src = in + g * 8;
    float* dst = out + g * 8;
    if (to_pair) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 hi, lo;
        sp_split8(v, hi, lo);
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 4) = lo;
    } else {
        float v[8];
        sp_unsplit8(*reinterpret_cast<const uint4*>(src), *reinterpret_cast<const uint4*>(src + 4), v);
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

extern "C" int al3d_sp_rows_convert_f16x3(const float* in, int64_t n, int channels, int to_pair, float* out, void* stream)
{
    AL3D_REQUIRE(n >= 0 && channels >= 8 && channels % 8 == 0, "al3d_sp_rows_convert_f16x3: channels must be a multiple of 8");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(in && out, "al3d_sp_rows_convert_f16x3: null pointer");
    const int64_t groups = n * (channels / 8);
    hipLaunchKernelGGL(sp_rows_convert_kernel, dim3((unsigned)al3d_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       groups, to_pair, out);
    AL3D_CHECK_LAUNCH("sp_rows_convert_kernel");
    return AL3D_OK;
}
