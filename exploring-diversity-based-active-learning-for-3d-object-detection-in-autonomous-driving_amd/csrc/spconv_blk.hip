// Submanifold sparse convolution, f16x3 arithmetic, BLOCK-STAGED gather (gfx950) -- levels 1-3 of the encoder.
//
// The per-tap kernels (spconv_glds.hip, spconv_wave.hip) fetch a neighbour row once per tap that references it --
// ~16 times per row on lidar data -- and the range kernel (spconv_rng.hip) once per (kz, ky) group: ~10 times.
// Here a workgroup owns a CHUNK of R consecutive output rows (R / 32 tiles) and stages the UNION of their 27-tap
// neighbourhoods ONCE into LDS (as pair rows: the f16 split happens once per staged row, not once per gathered
// (row, tap)); all 27 taps then read their A fragments from LDS through a 16-bit local index per (tap, row).
// With the level's rows in blocked order (al3d_sp_down_sites_blocked: 8 x 8 columns over all z, so that a chunk
// is a compact patch and its z halo is inside the block) the union is 1.5-1.8 x the chunk's rows.
//
// The plan (al3d_sp_block_plan: per chunk the staged row list + the local indices) depends on the level's table
// only and is shared by the level's SubM layers.  It is built without a sort or a hash: a chunk's neighbour ids
// cluster in a few id windows (its own strip of blocks, the strips above and below); up to 8 windows of 512 ids
// are opened greedily at the smallest uncovered id, a bitmap per window marks the referenced rows, local index =
// prefix popcount.  Deterministic, ascending; chunks that do not fit (more windows, or more than CAP rows: rows in
// an order without locality) are flagged and take a per-tap gather inside the same kernel (correct, slower).
//
// Workgroup: R / 32 consumer waves (one 32-row tile x all output channels each) + one producer wave streaming
// weight slabs (SU units of (tap, 16 input channels)) through a two-buffer LDS ring by LDS-DMA, one raw s_barrier
// per slab -- the structure of spconv_glds.hip / spconv_rng.hip, whose weight image (al3d_sp_pack_glds_f16x3) is
// used as it is.  Arithmetic and summation order are sp_conv_wave2's (taps ascending, channel units ascending,
// al * wd, ah * wl, ah * wh into one fp32 accumulator): BIT-IDENTICAL outputs whatever the row order (a row
// without a neighbour under a live tap adds +0, which never changes an accumulator that started at +0).
// Reference semantics: bevfusion/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361 (indice conv),
// geometry.h:248-298 (submanifold rulebook).
#include "glds_common.h"
#include "sp_rows.h"
#include "../../include/al3d.h"

#define BLK_NWIN 8
#define BLK_WSPAN 512
#define BLK_WWORDS (BLK_WSPAN / 32)
#define BLK_NONE 0xffffu

// ---------------------------------------------------------------------------------------------- plan
// one workgroup of R threads per chunk; thread r owns row chunk * R + r and its 27 table entries
template <int R>
__global__ __launch_bounds__(R) void sp_block_plan_kernel(const int* __restrict__ nbr, int64_t pitch, int n, int cap,
                                                         int2* __restrict__ hdr, int* __restrict__ rows,
                                                         unsigned short* __restrict__ loc)
{
    __shared__ int wbase[BLK_NWIN];
    __shared__ unsigned bits[BLK_NWIN * BLK_WWORDS];
    __shared__ int wpre[BLK_NWIN * BLK_WWORDS];
    __shared__ int red, over, wtot[4];
    const int c = blockIdx.x, r = threadIdx.x;
    const int64_t row = (int64_t)c * R + r;
    int v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = row < n ? nbr[(int64_t)k * pitch + row] : -1;
    for (int i = r; i < BLK_NWIN * BLK_WWORDS; i += R) bits[i] = 0u;
    if (r == 0) over = 0;
    int cur_lo = 0, nwin = 0;
    for (int w = 0; w < BLK_NWIN; ++w) {
        if (r == 0) red = 0x7fffffff;
        __syncthreads();
        int m = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 27; ++k) m = (v[k] >= cur_lo && v[k] < m) ? v[k] : m;
        if (m != 0x7fffffff) atomicMin(&red, m);
        __syncthreads();
        const int M = red;
        __syncthreads();
        if (M == 0x7fffffff) break;
        const int base = M & ~31;
        if (r == 0) wbase[w] = base;
        cur_lo = base + BLK_WSPAN;
        nwin = w + 1;
    }
    {   // anything beyond the last window?
        bool left = false;
#pragma unroll
        for (int k = 0; k < 27; ++k) left |= v[k] >= cur_lo;
        if (left && nwin == BLK_NWIN) over = 1;
    }
    __syncthreads();
    int wi[27];                                                             // window of each entry (-1: none)
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int w = -1;
        if (v[k] >= 0) {
            for (int j = 0; j < nwin; ++j)
                if (v[k] >= wbase[j] && v[k] < wbase[j] + BLK_WSPAN) w = j;
            if (w >= 0) atomicOr(&bits[w * BLK_WWORDS + ((v[k] - wbase[w]) >> 5)], 1u << (v[k] & 31));
        }
        wi[k] = w;
    }
    __syncthreads();
    // exclusive prefix of the popcounts over the NWIN * WWORDS = 128 words (R = 64: two words per thread)
    constexpr int WPT = BLK_NWIN * BLK_WWORDS / R;
    static_assert(WPT >= 1 && WPT * R == BLK_NWIN * BLK_WWORDS, "R must divide the word count");
    int cnt[WPT], tot = 0;
#pragma unroll
    for (int i = 0; i < WPT; ++i) { cnt[i] = __popc(bits[r * WPT + i]); tot += cnt[i]; }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((r & 63) >= d) incl += o;
    }
    if ((r & 63) == 63) wtot[r >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int i = 0; i < (r >> 6); ++i) wave_off += wtot[i];
    int excl = wave_off + incl - tot;
    int U = 0;
    for (int i = 0; i < (R + 63) / 64; ++i) U += wtot[i];
#pragma unroll
    for (int i = 0; i < WPT; ++i) { wpre[r * WPT + i] = excl; excl += cnt[i]; }
    __syncthreads();
    const bool fallback = over || U > cap;
    if (r == 0) hdr[c] = make_int2(fallback ? 0 : U, fallback ? 1 : 0);
    if (fallback) return;                                                   // uniform
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        unsigned l = BLK_NONE;
        if (wi[k] >= 0) {
            const int word = wi[k] * BLK_WWORDS + ((v[k] - wbase[wi[k]]) >> 5);
            l = (unsigned)(wpre[word] + __popc(bits[word] & ((1u << (v[k] & 31)) - 1u)));
        }
        loc[((int64_t)c * 27 + k) * R + r] = (unsigned short)l;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int word = r * WPT + i;
        unsigned m = bits[word];
        int pos = wpre[word];
        const int id0 = wbase[word / BLK_WWORDS] + 32 * (word % BLK_WWORDS);
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            rows[(int64_t)c * cap + pos++] = id0 + b;
        }
    }
}

// ---------------------------------------------------------------------------------------------- kernel
template <int CIN, int COUT, int R, int CAP, int SU, int NB>
struct BlkCfg {
    static constexpr int KG = CIN / 16;                   // 16-channel units per tap
    static constexpr int TN = COUT / 32;
    static constexpr int NROWS = TN * 32;
    static constexpr int UNIT_BYTES = 2 * NROWS * 32;     // two planes, 32-byte rows (the glds weight image)
    static constexpr int NWC = R / 32;                    // consumer waves: one tile each
    static constexpr int NWT = NWC + 1;
    static constexpr int NU = 27 * KG;
    static constexpr int NSLAB = NU / SU;
    static constexpr int SLAB_BYTES = SU * UNIT_BYTES;
    static constexpr int SLAB_PIECES = SLAB_BYTES / 1024;
    static constexpr int W_BYTES = NB * SLAB_BYTES;       // ring of NB slabs: the producer runs NB - 1 slabs ahead
    static constexpr int ROWB = CIN * 4;
    static constexpr int A_BYTES = CAP * ROWB;
    static constexpr int LOC_BYTES = 27 * R * 2;
    static constexpr int ZERO_OFF = W_BYTES + A_BYTES + LOC_BYTES;
    static constexpr int SMEM_BYTES = ZERO_OFF + 64;
    static constexpr int LPR = CIN / 8;                   // lanes per staged row: one 8-channel group (32 bytes) each
    static constexpr int RPI = 64 / LPR;                  // rows per wave and staging iteration
    static constexpr int EP_PITCH = 40;
    static constexpr int TPS = SU >= KG ? SU / KG : 1;    // whole taps per slab ...
    static constexpr int SPT = SU >= KG ? 1 : KG / SU;    // ... or slabs per tap
    static_assert(CIN % 32 == 0 && COUT % 32 == 0 && R % 32 == 0 && NU % SU == 0, "shape");
    static_assert(SU >= KG ? SU % KG == 0 : KG % SU == 0, "a slab is whole taps or a whole fraction of one");
    static_assert(NB >= 2 && (NB - 1) * SLAB_PIECES <= 63, "vmcnt is a 6-bit counter");
    static_assert(A_BYTES >= NWC * 32 * EP_PITCH * 4, "the epilogue transposes through the staged rows' storage");
    static_assert(A_BYTES >= NWC * 32 * ROWB, "the per-tap path stages 32 rows per wave");
    static_assert(W_BYTES + NROWS * 32 + 3072 < 65536, "ds_read immediate offsets are 16 bits");
    static_assert(LOC_BYTES % 16 == 0 && SMEM_BYTES <= 160 * 1024, "LDS budget");
    static_assert(CAP < 0xffff, "16-bit local indices");
};

// position of 16-byte chunk q of staged row j: q ^ swz(j) -- conflict-free for the b128 lane groups whenever the 32
// rows of a fragment sit at consecutive staged positions (128-byte rows: as spconv_rng.hip; longer rows start at bank 0)
template <int ROWB> __device__ __forceinline__ unsigned blk_swz(unsigned j)
{
    if constexpr (ROWB == 128) return (j >> 1) & 7u;
    else return j & 15u;
}

__device__ __forceinline__ void blk_lds_read_u16(unsigned& d, unsigned addr)
{
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
}

// One pipelined unit (as spconv_rng.hip): the ds_reads of the NEXT unit's A / B fragments interleaved with the MFMAs
// of the current unit, closed by the lgkmcnt wait -- one asm block, so no register is visible to hipcc in flight.
template <int TN, int OFF, int PL>
__device__ __forceinline__ void blk_read_next_mfma(gl_f32x4& nlo, gl_f32x4& nhi, f16x8 (&nwh)[TN], f16x8 (&nwl)[TN],
                                                   f32x16 (&acc)[TN], const f16x8& al, const f16x8& ah,
                                                   const f16x8 (&wd)[TN], const f16x8 (&wl)[TN], const f16x8 (&wh)[TN],
                                                   unsigned a0, unsigned a1, unsigned b)
{
    static_assert(TN == 1 || TN == 2, "tile counts of the pipelined channel pairs");
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

template <int CIN, int COUT, int R, int CAP, int SU, int NB>
__global__ __launch_bounds__(64 * (R / 32 + 1)) void sp_conv_blk_kernel(const float* __restrict__ fin,
                                                                       const int* __restrict__ nbr, int pitch,
                                                                       const unsigned* __restrict__ tmask, int ntiles,
                                                                       const int2* __restrict__ hdr,
                                                                       const int* __restrict__ prow,
                                                                       const unsigned short* __restrict__ ploc,
                                                                       const unsigned char* __restrict__ wpk,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift,
                                                                       const float* __restrict__ residual, int relu,
                                                                       float* __restrict__ fout, int n_out, int io, int abl)
{
    using C = BlkCfg<CIN, COUT, R, CAP, SU, NB>;
    constexpr int KG = C::KG, TN = C::TN, NROWS = C::NROWS, UNIT_BYTES = C::UNIT_BYTES, NWC = C::NWC;
    constexpr int ROWB = C::ROWB, LPR = C::LPR, RPI = C::RPI, NSLAB = C::NSLAB, TPS = C::TPS, SPT = C::SPT;
    constexpr bool PIPE = TN <= 2;                                           // software-pipelined units (asm for TN 1, 2)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware placement: each XCD gets one contiguous range of chunks
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const bool consumer = wave < NWC;

    const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;
    const unsigned a_base = smem_base + C::W_BYTES;
    const unsigned loc_base = a_base + C::A_BYTES;
    const unsigned zero_base = smem_base + C::ZERO_OFF;

    const int2 h = hdr[chunk];
    const int U = (abl & 2) ? 0 : __builtin_amdgcn_readfirstlane(h.x);
    const bool pertap = __builtin_amdgcn_readfirstlane(h.y) != 0;

    // the producer's first NB - 1 slabs go out before anything else (a slab past the end re-fetches an earlier one
    // into a free buffer: the count of requests per slab is what the counted waits rely on)
    auto issue_slab = [&](int s) {
        const int sc = s < NSLAB ? s : s - NSLAB;
        const unsigned char* src = wpk + (size_t)sc * C::SLAB_BYTES + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + (s % NB) * C::SLAB_BYTES);
        gl_static_for<C::SLAB_PIECES>([&](auto PC) {
            constexpr int pc = decltype(PC)::value;
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + pc * 1024), (lds_void*)(size_t)(dst + pc * 1024), 16, 0, 0);
        });
    };
    if (!consumer) {
#pragma unroll
        for (int s = 0; s < NB - 1; ++s) issue_slab(s);
    }

    if (tid < 16) reinterpret_cast<float*>(smem + C::ZERO_OFF)[tid] = 0.f;
    if (!pertap && consumer) {
        // ---- local indices of the chunk: 27 x R x 2 bytes, contiguous in the plan
        const uint4* src = reinterpret_cast<const uint4*>(ploc + (int64_t)chunk * 27 * R);
        for (int i = tid; i < C::LOC_BYTES / 16; i += 64 * NWC)
            *reinterpret_cast<uint4*>(smem + C::W_BYTES + C::A_BYTES + i * 16) = src[i];
        // ---- the staged rows: lane group g of LPR lanes fetches row j's eight-channel group (32 bytes), splits it
        // (f32 rows) and stores its two 16-byte planes at the swizzled positions of staged row j.  Two iterations'
        // loads are requested before the first one's stores.
        const int g = lane % LPR, jr = lane / LPR;
        const int* ids = prow + (int64_t)chunk * CAP;
        auto put = [&](int j, const uint4& a, const uint4& b) {
            uint4 hi = a, lo = b;
            if (!(io & SP_IO_IN_PAIR)) {
                const float v[8] = {__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, a.z),
                                    __builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y),
                                    __builtin_bit_cast(float, b.z), __builtin_bit_cast(float, b.w)};
                sp_split8(v, hi, lo);
            }
            const unsigned sw = blk_swz<ROWB>((unsigned)j);
            unsigned char* rowp = smem + C::W_BYTES + j * ROWB;
            *reinterpret_cast<uint4*>(rowp + (((2 * g) ^ sw) << 4)) = hi;
            *reinterpret_cast<uint4*>(rowp + (((2 * g + 1) ^ sw) << 4)) = lo;
        };
        constexpr int UNR = 4;
        for (int j0 = wave * RPI; j0 < U; j0 += UNR * NWC * RPI) {
            int id[UNR];
            uint4 a[UNR], b[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const int j = j0 + t * NWC * RPI + jr;
                id[t] = ids[j < U ? j : U - 1];
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const float* p = fin + (int64_t)id[t] * CIN + g * 8;
                a[t] = *reinterpret_cast<const uint4*>(p);
                b[t] = *reinterpret_cast<const uint4*>(p + 4);
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const int j = j0 + t * NWC * RPI + jr;
                if (j < U) put(j, a[t], b[t]);
            }
        }
    }

    if (!consumer) {
        // ================= producer wave: NB - 1 slabs ahead of the consumers
        for (int s = 0; s < NSLAB; ++s) {
            gl_wait_vm<(NB - 2) * C::SLAB_PIECES>();                         // slab s has landed
            if (!(abl & 1) || s == 0) __builtin_amdgcn_s_barrier();          // ... and slab s-1's buffer is free
            issue_slab(s + NB - 1);
        }
        gl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                                        // the consumers' hand-over to their epilogue
        return;
    }

    // ================= consumer waves: one tile each
    const int tile = chunk * NWC + wave;
    const unsigned wmask = __builtin_amdgcn_readfirstlane(tile < ntiles ? tmask[tile] : 0u);
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const unsigned offB = smem_base + (2 * fr + (fh ^ ((fr >> 3) & 1))) * 16;  // the weight image's swizzle
    const unsigned priv = a_base + wave * (32 * ROWB);                       // per-tap path: this wave's 32 rows
    const unsigned loc_lane = loc_base + (wave * 32 + fr) * 2;
    __builtin_amdgcn_s_waitcnt(0);                                           // this wave's staging stores

    // the per-tap path's gather of one tap's 32 neighbour rows into the wave's private rows
    auto gather_tap = [&](int k, unsigned& rowa, unsigned& sw, bool& ok) {
        const int g = lane % LPR, jr = lane / LPR;
#pragma unroll
        for (int j0 = 0; j0 < 32; j0 += RPI) {
            const int j = j0 + jr;
            const int id = nbr[(int64_t)k * pitch + (int64_t)tile * 32 + j];
            if (id >= 0) {
                const float* p = fin + (int64_t)id * CIN + g * 8;
                const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 4);
                uint4 hi = a, lo = b;
                if (!(io & SP_IO_IN_PAIR)) {
                    const float v[8] = {__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y),
                                        __builtin_bit_cast(float, a.z), __builtin_bit_cast(float, a.w),
                                        __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y),
                                        __builtin_bit_cast(float, b.z), __builtin_bit_cast(float, b.w)};
                    sp_split8(v, hi, lo);
                }
                const unsigned swj = blk_swz<ROWB>((unsigned)j);
                unsigned char* rowp = smem + C::W_BYTES + wave * (32 * ROWB) + j * ROWB;
                *reinterpret_cast<uint4*>(rowp + (((2 * g) ^ swj) << 4)) = hi;
                *reinterpret_cast<uint4*>(rowp + (((2 * g + 1) ^ swj) << 4)) = lo;
            }
        }
        const int idf = nbr[(int64_t)k * pitch + (int64_t)tile * 32 + fr];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        ok = idf >= 0;
        rowa = priv + fr * ROWB;
        sw = blk_swz<ROWB>((unsigned)fr);
    };
    auto mfma3 = [&](const f16x8& ah, const f16x8& al, const f16x8 (&wh)[TN], const f16x8 (&wl)[TN]) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const f16x8 wd = gl_lift_down(wh[jn]);
            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd, acc[jn], 0, 0, 0);              // smallest first
            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[jn], acc[jn], 0, 0, 0);
            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[jn], acc[jn], 0, 0, 0);
        }
    };

    // staged rows of the current tap(s): row address, swizzle term, presence -- per tap of the slab
    unsigned rowa[TPS], sw[TPS];
    bool ok[TPS];
#pragma unroll
    for (int t = 0; t < TPS; ++t) { rowa[t] = zero_base; sw[t] = 0u; ok[t] = false; }

    for (int s = 0; s < NSLAB; ++s) {
        const int k0 = SU >= KG ? s * TPS : s / SPT;                         // first (or only) tap of the slab
        const int kgb = SU >= KG ? 0 : (s % SPT) * SU;                       // its first unit within the tap
        unsigned live = 0u;
#pragma unroll
        for (int t = 0; t < TPS; ++t) live |= ((wmask >> (k0 + t)) & 1u) << t;
        if (!pertap && (SU >= KG || kgb == 0)) {
            // the slab's local indices (static data: read before the barrier)
#pragma unroll
            for (int t = 0; t < TPS; ++t) {
                unsigned lid;
                blk_lds_read_u16(lid, loc_lane + (k0 + t) * (R * 2));
                ok[t] = lid != BLK_NONE;
                rowa[t] = a_base + lid * ROWB;
                sw[t] = blk_swz<ROWB>(lid);
            }
        }
        if (!(abl & 1) || s == 0) __builtin_amdgcn_s_barrier();
        if (!live || (abl & 4)) continue;                                    // wave-uniform: no neighbour under these taps
        const unsigned bslab = offB + (s % NB) * C::SLAB_BYTES;
        auto a_addr = [&](int t, int kg, int hl) -> unsigned {
            return ok[t] ? rowa[t] + (((unsigned)(4 * kg + 2 * fh + hl) ^ sw[t]) << 4) : zero_base;
        };
        if (pertap || !PIPE) {
            gl_static_for<SU>([&](auto UU) {
                constexpr int u = decltype(UU)::value;
                constexpr int t = SU >= KG ? u / KG : 0;
                const int kg = SU >= KG ? u % KG : kgb + u;
                if (!((live >> t) & 1u)) return;
                if (pertap && kg == 0) gather_tap(k0 + t, rowa[t], sw[t], ok[t]);
                gl_f32x4 vh, vl;
                f16x8 wh[TN], wl[TN];
                gl_lds_read_ab<TN, u * UNIT_BYTES, NROWS * 32>(vh, vl, wh, wl, a_addr(t, kg, 0), a_addr(t, kg, 1), bslab);
                mfma3(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, vl), wh, wl);
                if (pertap && kg == KG - 1) __builtin_amdgcn_wave_barrier();
            });
        } else if constexpr (PIPE) {
            // all SU units of the slab, unit u + 1's fragments read while unit u's products run (a tap without a
            // neighbour in this tile reads the zero row: +0 products)
            gl_f32x4 vh[2], vl[2];
            f16x8 wh[2][TN], wl[2][TN];
            gl_lds_read_ab<TN, 0, NROWS * 32>(vh[0], vl[0], wh[0], wl[0], a_addr(0, kgb, 0), a_addr(0, kgb, 1), bslab);
            gl_static_for<SU>([&](auto UU) {
                constexpr int u = decltype(UU)::value;
                constexpr int cur = u & 1, nxt = cur ^ 1;
                const f16x8 ah = __builtin_bit_cast(f16x8, vh[cur]), al = __builtin_bit_cast(f16x8, vl[cur]);
                if constexpr (u + 1 < SU) {
                    constexpr int tn = SU >= KG ? (u + 1) / KG : 0;
                    const int kgn = SU >= KG ? (u + 1) % KG : kgb + u + 1;
                    f16x8 wd[TN];
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) wd[jn] = gl_lift_down(wh[cur][jn]);
                    blk_read_next_mfma<TN, (u + 1) * UNIT_BYTES, NROWS * 32>(vh[nxt], vl[nxt], wh[nxt], wl[nxt], acc, al, ah, wd,
                                                                             wl[cur], wh[cur], a_addr(tn, kgn, 0),
                                                                             a_addr(tn, kgn, 1), bslab);
                } else {
                    mfma3(ah, al, wh[cur], wl[cur]);
                }
            });
        }
    }
    __builtin_amdgcn_s_barrier();                                            // every wave is done with the staged rows

    // ---- epilogue (as sp_conv_glds_kernel): transpose each 32 x 32 C tile through LDS, BN / residual / ReLU / store
    float* scr = reinterpret_cast<float*>(smem + C::W_BYTES + wave * (32 * C::EP_PITCH * 4));
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

// ---------------------------------------------------------------------------------------------- C ABI
// the shapes the conv kernel is built for: R rows per chunk, CAP staged rows (two workgroups per CU)
static bool blk_shape(int cin, int cout, int* R, int* cap)
{
    if (cin == 32 && cout == 32) { *R = 128; *cap = 288; return true; }
    if (cin == 64 && cout == 64) { *R = 128; *cap = 208; return true; }
    if (cin == 128 && cout == 128) { *R = 64; *cap = 104; return true; }
    return false;
}

extern "C" int al3d_sp_block_shape(int cin, int cout, int* rows_per_chunk, int* staged_cap)
{
    AL3D_REQUIRE(rows_per_chunk && staged_cap, "al3d_sp_block_shape: null pointer");
    if (!blk_shape(cin, cout, rows_per_chunk, staged_cap)) {
        *rows_per_chunk = 0; *staged_cap = 0;
        return al3d_fail(AL3D_EINVAL, "al3d_sp_block_shape: no block-staged kernel for Cin=%d Cout=%d", cin, cout);
    }
    return AL3D_OK;
}

extern "C" int al3d_sp_block_plan(const int* nbr, int64_t nbr_pitch, int K, int n_out, int rows_per_chunk, int staged_cap,
                                  int* out_hdr, int* out_rows, void* out_loc, void* stream)
{
    AL3D_REQUIRE(K == 27 && n_out >= 0 && nbr_pitch >= n_out, "al3d_sp_block_plan: 27-tap tables only");
    AL3D_REQUIRE(rows_per_chunk == 64 || rows_per_chunk == 128, "al3d_sp_block_plan: 64 or 128 rows per chunk");
    AL3D_REQUIRE(staged_cap > 0 && staged_cap < 0xffff, "al3d_sp_block_plan: bad cap");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(nbr && out_hdr && out_rows && out_loc, "al3d_sp_block_plan: null pointer");
    AL3D_REQUIRE(nbr_pitch % 256 == 0, "al3d_sp_block_plan: nbr_pitch must be al3d_sp_table_pitch(n_out)");
    const int nchunks = (int)al3d_cdiv(n_out, rows_per_chunk);
    hipStream_t s = (hipStream_t)stream;
    if (rows_per_chunk == 128)
        hipLaunchKernelGGL((sp_block_plan_kernel<128>), dim3((unsigned)nchunks), dim3(128), 0, s, nbr, nbr_pitch, n_out,
                           staged_cap, (int2*)out_hdr, out_rows, (unsigned short*)out_loc);
    else
        hipLaunchKernelGGL((sp_block_plan_kernel<64>), dim3((unsigned)nchunks), dim3(64), 0, s, nbr, nbr_pitch, n_out,
                           staged_cap, (int2*)out_hdr, out_rows, (unsigned short*)out_loc);
    AL3D_CHECK_LAUNCH("sp_block_plan_kernel");
    return AL3D_OK;
}

#define BLK_DISPATCH(CI, CO, RR, CAPV, SUV, NBV)                                                                     \
    if (cin == CI && cout == CO) {                                                                              \
        const int nchunks = (int)al3d_cdiv(n_out, RR);                                                          \
        hipLaunchKernelGGL((sp_conv_blk_kernel<CI, CO, RR, CAPV, SUV, NBV>), dim3((unsigned)nchunks),               \
                           dim3(64 * (RR / 32 + 1)), 0, s, fin, nbr, nbr_pitch, tile_mask, ntiles, (const int2*)plan_hdr, \
                           plan_rows, (const unsigned short*)plan_loc, (const unsigned char*)wgt_image, scale, shift, \
                           residual, relu, fout, n_out, io, abl);                                               \
        AL3D_CHECK_LAUNCH("sp_conv_blk_kernel");                                                                \
        return AL3D_OK;                                                                                         \
    }

extern "C" int al3d_sp_conv_blk_f16x3(const float* fin, const int* nbr, int nbr_pitch, const unsigned* tile_mask,
                                      const int* plan_hdr, const int* plan_rows, const void* plan_loc, int K,
                                      const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                                      const float* residual, int relu, float* fout, int n_out, int io, void* stream)
{
    AL3D_REQUIRE(K == 27 && n_out >= 0, "al3d_sp_conv_blk_f16x3: 27-tap submanifold layers only");
    AL3D_REQUIRE(io >= 0 && io < 8, "al3d_sp_conv_blk_f16x3: bad io flags");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && tile_mask && plan_hdr && plan_rows && plan_loc && wgt_image && fout,
                 "al3d_sp_conv_blk_f16x3: null pointer");
    AL3D_REQUIRE(scale, "al3d_sp_conv_blk_f16x3: scale carries the weight exponent and is required");
    AL3D_REQUIRE(nbr_pitch >= n_out && nbr_pitch % 256 == 0, "al3d_sp_conv_blk_f16x3: nbr_pitch must be al3d_sp_table_pitch(n_out)");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)al3d_cdiv(n_out, 32);
    int abl = 0;
#ifdef AL3D_BLK_ABLATE
    if (const char* e = getenv("AL3D_BLK_ABL")) abl = atoi(e);              // tuning build only (make EXTRA=-DAL3D_BLK_ABLATE)
#endif
    BLK_DISPATCH(32, 32, 128, 288, 6, 3)
    BLK_DISPATCH(64, 64, 128, 208, 2, 2)
    BLK_DISPATCH(128, 128, 64, 104, 1, 2)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_blk_f16x3: no kernel for Cin=%d Cout=%d", cin, cout);
}
