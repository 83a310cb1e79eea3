// Level-0 sparse convolutions (16 input channels) on RASTER-ordered rows, f16x3 arithmetic (gfx950).
//
// The five 16 -> 16 submanifold layers and the 16 -> 32 strided layer of the CBGS middle encoder
// (det3d/models/backbones/scn.py:331-347; rulebook semantics bevfusion/mmdet3d/ops/spconv/include/spconv/
// geometry.h:248-298) read the voxelizer's rows.  A 16-channel row is 64 bytes, and the register-gather kernel
// (spconv_wave.hip) fetches it as fragment-shaped pieces: 64 lanes -> 64 different (row, 16-byte) addresses per load
// instruction, 128 texture-address lookups per (32-row tile, tap) against 96 cycles of products -- the kernel is bound
// by the CU's address unit (2 instructions x 64 lanes x 9.9 live taps x 240k tiles / 256 CUs = 1.19 M cycles of the
// 745 us a layer takes).  The voxelizer hands its rows over in first-appearance order (the reference's contract for
// example["coordinates"]); inside the encoder the order of a level's rows is free (the dense scatter at the end is
// order-invariant), so this file
//   1. renumbers the level-0 rows in raster order (b, z, y, x) -- al3d_sp_raster_perm: a counting sort over the
//      (b, z, y) lines + a bit-mask rank inside each line, no global sort --,
//   2. describes every (32-row tile, (kz, ky) group) with a live tap as one ITEM (lo, len): in raster order the
//      neighbours of 32 consecutive rows under the three kx taps of a group lie in one short contiguous index range
//      (median 33 rows, <= 48 for 96 %) -- al3d_sp_tile_items, a flat list in (tile, group) order --,
//   3. runs the layer as a stream of items per wave (sp_conv_r16_kernel): a range is CONTIGUOUS memory, so plain
//      16-byte-per-lane loads fetch it fully coalesced (three 1 KiB requests per item instead of 3 x 128 lane lookups);
//      the rows wait in registers P items ahead, are written to the wave's LDS slot when the item is consumed, and the
//      three taps read their A fragments from it at LDS row (index - lo); ALL 27 taps' weights stay resident in LDS
//      (27 KB at 16 output channels), so the main loop has no barrier and no producer wave: waves are independent and
//      stream their items across tile boundaries.  Ranges longer than CAP rows (tiles straddling a z plane) fall back
//      to a per-row gather inside the same item.
// Arithmetic and summation order are sp_conv_wave2's (tap ascending; per tap xl' wd, xh wl, xh wh into one fp32
// accumulator of v_mfma_f32_32x32x16_f16): outputs are bit-identical to it row for row.
#include "glds_common.h"
#include "sp_rows.h"
#include "al3d_scan.h"

// =====================================================================================================
// 1. raster permutation
// =====================================================================================================
struct L0Dims { int B, D, H, W; };

__global__ __launch_bounds__(256) void l0_line_count_kernel(const int* __restrict__ coords, int n, L0Dims g,
                                                            int* __restrict__ cnt, int* __restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
    const int line = (c.x * g.D + c.y) * g.H + c.z;
    slot[i] = atomicAdd(&cnt[line], 1);
}

__global__ __launch_bounds__(256) void l0_bucket_kernel(const int* __restrict__ coords, int n, L0Dims g,
                                                        const int* __restrict__ base, const int* __restrict__ slot,
                                                        unsigned short* __restrict__ bx, int* __restrict__ bid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
    const int line = (c.x * g.D + c.y) * g.H + c.z;
    const int p = base[line] + slot[i];
    bx[p] = (unsigned short)c.w;
    bid[p] = i;
}

// One wave per 64 line slots; every non-empty line is ranked by the whole wave: its members set their x bit in a
// W-bit mask in LDS (W <= 2048: one word per lane), a prefix popcount over the words gives each member its rank.
__global__ __launch_bounds__(256) void l0_rank_kernel(const int* __restrict__ cnt, const int* __restrict__ base, int L,
                                                      L0Dims g, const unsigned short* __restrict__ bx,
                                                      const int* __restrict__ bid, int* __restrict__ perm,
                                                      int* __restrict__ coords_r)
{
    __shared__ unsigned s_bits[4][64];
    __shared__ int s_pre[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int line0 = (blockIdx.x * 4 + wave) * 64;
    const int myline = line0 + lane;
    const int c = myline < L ? cnt[myline] : 0;
    const int b0 = myline < L ? base[myline] : 0;
    unsigned long long live = __ballot(c > 0);
    while (live) {
        const int l = __builtin_ctzll(live);
        live &= live - 1ull;
        const int cc = __builtin_amdgcn_readlane(c, l), bb = __builtin_amdgcn_readlane(b0, l);
        const int line = line0 + l;
        const int y = line % g.H, z = (line / g.H) % g.D, b = line / (g.H * g.D);
        s_bits[wave][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < cc; m += 64) {
            const unsigned x = bx[bb + m];
            atomicOr(&s_bits[wave][x >> 5], 1u << (x & 31u));
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int pc = __popc(s_bits[wave][lane]);
            int inc = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off);
                if (lane >= off) inc += t;
            }
            s_pre[wave][lane] = inc - pc;
        }
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < cc; m += 64) {
            const unsigned x = bx[bb + m];
            const int id = bid[bb + m];
            const int rank = s_pre[wave][x >> 5] + __popc(s_bits[wave][x >> 5] & ((1u << (x & 31u)) - 1u));
            const int pos = bb + rank;
            perm[pos] = id;
            *reinterpret_cast<int4*>(coords_r + 4 * (int64_t)pos) = make_int4(b, z, y, (int)x);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the same permutation for frame-sorted inputs, one workgroup per frame, everything but the bucket arrays in LDS.
// The voxelizer hands over the frames of a batch one after the other (coords[:, 0] ascending) with at most 65,535 rows
// each; a frame's (z, y) lines then fit 16-bit counters in LDS (D H <= 43,008: 84 KB) and the three global passes above
// (7.68 M returning atomics on a 21 MB counter array at the memory side, a 5.4 M-entry scan, one wave per 64 line slots
// with two dependent global round trips per non-empty line: 0.95 ms per 128-frame batch) become LDS atomics, an LDS scan
// and an LDS bit-mask rank over chunks of 512 lines.
#define L0F_NT 1024
#define L0F_MAXLINES 43008
#define L0F_CHUNK 512
#define L0F_K 4
// The caller's promise (rows frame-sorted, at most 65,535 per frame) is CHECKED, not trusted: this pass sets
// status bit 0 when coords[:, 0] is not ascending within [0, B); the sort kernel then writes the identity permutation
// (every row exactly once, valid coordinates -- correct, unsorted) instead of walking 16-bit counters that would wrap, and a
// frame beyond the 16-bit limit does the same for its own rows and sets bit 1.  The host raises on a non-zero status.
__global__ __launch_bounds__(256) void l0_check_frames_kernel(const int* __restrict__ coords, int n, int B, int* __restrict__ status)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = coords[4 * (int64_t)i];
    const int bn = i + 1 < n ? coords[4 * (int64_t)(i + 1)] : b;
    if (b < 0 || b >= B || bn < b) atomicOr(status, 1);
}

__global__ __launch_bounds__(L0F_NT) void l0_frame_sort_kernel(const int* __restrict__ coords, int n, L0Dims g,
                                                               unsigned short* __restrict__ slot_ws,
                                                               unsigned* __restrict__ bkey, int* __restrict__ bid,
                                                               int* __restrict__ perm, int* __restrict__ coords_r,
                                                               int* __restrict__ status)
{
    __shared__ unsigned cw[L0F_MAXLINES / 2 + 2];              // packed 16-bit line counters, then exclusive prefixes
    __shared__ unsigned bits[L0F_CHUNK * 32];                  // x bit masks of one chunk of lines
    __shared__ int s_part[L0F_NT / 64];
    __shared__ int s_lo, s_hi;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int LH = g.D * g.H;
    if (tid < 2) {                                             // frame range: lower bounds of b and b + 1 in coords[:, 0]
        const int key = b + tid;
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (coords[4 * (int64_t)mid] < key) lo = mid + 1; else hi = mid;
        }
        if (tid == 0) s_lo = lo; else s_hi = lo;
    }
    for (int i = tid; i < L0F_MAXLINES / 2 + 2; i += L0F_NT) cw[i] = 0u;
    __syncthreads();
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) {
        // not frame-sorted: the frame ranges mean nothing.  Identity over an equal slice of the rows per workgroup.
        const int64_t i0 = (int64_t)n * b / g.B, i1 = (int64_t)n * (b + 1) / g.B;
        for (int64_t i = i0 + tid; i < i1; i += L0F_NT) {
            perm[i] = (int)i;
            *reinterpret_cast<int4*>(coords_r + 4 * i) = *reinterpret_cast<const int4*>(coords + 4 * i);
        }
        return;
    }
    const int lo = s_lo, cnt_f = s_hi - s_lo;
    if (cnt_f <= 0) return;                                    // uniform
    if (cnt_f > 65535) {                                       // beyond the 16-bit counters: identity for this frame's rows
        if (tid == 0) atomicOr(status, 2);
        for (int i = tid; i < cnt_f; i += L0F_NT) {
            perm[lo + i] = lo + i;
            *reinterpret_cast<int4*>(coords_r + 4 * (int64_t)(lo + i)) = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)(lo + i));
        }
        return;
    }
    // A: arrival slot of every row in its line
    for (int i = tid; i < cnt_f; i += L0F_NT) {
        const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)(lo + i));
        const int line = c.y * g.H + c.z;
        const unsigned old = atomicAdd(&cw[line >> 1], (line & 1) ? 0x10000u : 1u);
        slot_ws[lo + i] = (unsigned short)((line & 1) ? old >> 16 : old & 0xffffu);
    }
    __syncthreads();
    // B: exclusive scan of the LH counters in place (42 consecutive lines = 21 words per thread)
    {
        constexpr int WPT = (L0F_MAXLINES / 2) / L0F_NT;      // 21
        unsigned w[WPT];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            w[k] = cw[tid * WPT + k];
            sum += (int)(w[k] & 0xffffu) + (int)(w[k] >> 16);
        }
        const int lane = tid & 63, wave = tid >> 6;
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        if (wave == 0) {
            int v = lane < L0F_NT / 64 ? s_part[lane] : 0;
#pragma unroll
            for (int off = 1; off < L0F_NT / 64; off <<= 1) {
                const int t = __shfl_up(v, off);
                if (lane >= off) v += t;
            }
            if (lane < L0F_NT / 64) s_part[lane] = v;          // inclusive wave totals
        }
        __syncthreads();
        int run = (wave > 0 ? s_part[wave - 1] : 0) + inc - sum;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int c0 = (int)(w[k] & 0xffffu), c1 = (int)(w[k] >> 16);
            cw[tid * WPT + k] = (unsigned)run | ((unsigned)(run + c0) << 16);
            run += c0 + c1;
        }
        if (tid == L0F_NT - 1) cw[L0F_MAXLINES / 2] = (unsigned)run;   // base[MAXLINES] = the frame's row count
    }
    __syncthreads();
    auto base_of = [&](int line) -> int { const unsigned v = cw[line >> 1]; return (int)((line & 1) ? v >> 16 : v & 0xffffu); };
    // C: rows grouped by line (arrival order inside a line)
    for (int i = tid; i < cnt_f; i += L0F_NT) {
        const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)(lo + i));
        const int line = c.y * g.H + c.z;
        const int p = lo + base_of(line) + slot_ws[lo + i];
        bkey[p] = ((unsigned)line << 10) | (unsigned)c.w;         // line < 43,008 (16 bits), x < 1,024
        bid[p] = lo + i;
    }
    __syncthreads();                                           // the bucket arrays are read back by other threads below
    // D: rank inside each line from an x bit mask, 512 lines at a time.  A chunk costs three barriers; its bucket entries
    // (the first L0F_K per thread) are fetched one chunk ahead and kept in registers for both phases, so the loop waits for
    // global memory once, not twice per chunk (0.43 -> 0.2 ms per 128-frame batch)
    auto chunk_end = [&](int L) -> int { return L + L0F_CHUNK < LH ? base_of(L + L0F_CHUNK) : cnt_f; };
    auto next_chunk = [&](int L) -> int {                      // first chunk at or after L that holds a row (uniform)
        while (L < LH && base_of(L) == chunk_end(L)) L += L0F_CHUNK;
        return L;
    };
    unsigned ck[L0F_K], nk[L0F_K];
    int cid[L0F_K], nid[L0F_K];
    auto fetch = [&](int L, unsigned (&k)[L0F_K], int (&id)[L0F_K]) {
        const int q0 = L < LH ? base_of(L) : 0, q1 = L < LH ? chunk_end(L) : 0;
#pragma unroll
        for (int e = 0; e < L0F_K; ++e) {
            const int p = q0 + tid + e * L0F_NT;
            const int pc = p < q1 ? p : 0;                     // unconditional loads (clamped): the prefetch stays in flight
            k[e] = bkey[lo + pc];
            id[e] = bid[lo + pc];
        }
    };
    int Lc = next_chunk(0);
    fetch(Lc, ck, cid);
    while (Lc < LH) {
        const int p0 = base_of(Lc), p1 = chunk_end(Lc);
        const int Ln = next_chunk(Lc + L0F_CHUNK);
        fetch(Ln, nk, nid);
        for (int i = tid; i < L0F_CHUNK * 32; i += L0F_NT) bits[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < L0F_K; ++e)
            if (p0 + tid + e * L0F_NT < p1) {
                const unsigned x = ck[e] & 1023u;
                atomicOr(&bits[((int)(ck[e] >> 10) - Lc) * 32 + (x >> 5)], 1u << (x & 31u));
            }
        for (int p = p0 + tid + L0F_K * L0F_NT; p < p1; p += L0F_NT) {       // dense chunks: the entries beyond the held ones
            const unsigned k = bkey[lo + p], x = k & 1023u;
            atomicOr(&bits[((int)(k >> 10) - Lc) * 32 + (x >> 5)], 1u << (x & 31u));
        }
        __syncthreads();
        auto emit = [&](unsigned k, int id) {
            const unsigned x = k & 1023u;
            const int line = (int)(k >> 10);
            const unsigned* m = &bits[(line - Lc) * 32];
            int rank = __popc(m[x >> 5] & ((1u << (x & 31u)) - 1u));
            for (unsigned w = 0; w < (x >> 5); ++w) rank += __popc(m[w]);
            const int pos = lo + base_of(line) + rank;
            perm[pos] = id;
            *reinterpret_cast<int4*>(coords_r + 4 * (int64_t)pos) = make_int4(b, line / g.H, line % g.H, (int)x);
        };
#pragma unroll
        for (int e = 0; e < L0F_K; ++e)
            if (p0 + tid + e * L0F_NT < p1) emit(ck[e], cid[e]);
        for (int p = p0 + tid + L0F_K * L0F_NT; p < p1; p += L0F_NT) emit(bkey[lo + p], bid[lo + p]);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < L0F_K; ++e) { ck[e] = nk[e]; cid[e] = nid[e]; }
        Lc = Ln;
    }
}

static inline int64_t l0_lines(int B, int D, int H) { return (int64_t)B * D * H; }

extern "C" int64_t al3d_sp_raster_perm_workspace_bytes(int n, int B, int D, int H)
{
    const int64_t L = l0_lines(B, D, H);
    return 256 + 2 * al3d_align(L * 4, 256) + al3d_scan_workspace_bytes(L) + al3d_align((int64_t)n * 4, 256) * 3 +
           al3d_align((int64_t)n * 2, 256);
}

extern "C" int al3d_sp_raster_perm(const int* coords, int n, int B, int D, int H, int W, int frame_rows_max,
                                   void* workspace, int* perm, int* coords_raster, void* stream)
{
    AL3D_REQUIRE(n >= 0 && B > 0 && D > 0 && H > 0 && W > 0 && W <= 2048, "al3d_sp_raster_perm: bad sizes (W <= 2048)");
    AL3D_REQUIRE(l0_lines(B, D, H) < (1ll << 31), "al3d_sp_raster_perm: too many (b, z, y) lines");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(coords && workspace && perm && coords_raster, "al3d_sp_raster_perm: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t L = l0_lines(B, D, H);
    char* w = (char*)workspace;
    int* status = (int*)w; w += 256;                           // workspace[0]: 0 = ok, else the violated promise (see header)
    if (hipMemsetAsync(status, 0, 4, s) != hipSuccess) return al3d_fail(AL3D_ELAUNCH, "al3d_sp_raster_perm: memset failed");
    int* cnt = (int*)w; w += al3d_align(L * 4, 256);
    int* base = (int*)w; w += al3d_align(L * 4, 256);
    void* scan_ws = w; w += al3d_scan_workspace_bytes(L);
    int* slot = (int*)w; w += al3d_align((int64_t)n * 4, 256);
    int* bid = (int*)w; w += al3d_align((int64_t)n * 4, 256);
    unsigned short* bx = (unsigned short*)w; w += al3d_align((int64_t)n * 2, 256);
    int* bl = (int*)w;
    const L0Dims g{B, D, H, W};
    if (frame_rows_max > 0 && frame_rows_max <= 65535 && D * H <= L0F_MAXLINES && W <= 1024) {
        // frame-sorted input (coords[:, 0] ascending, at most frame_rows_max rows per frame): one workgroup per frame
        unsigned short* slot16 = (unsigned short*)slot;
        hipLaunchKernelGGL(l0_check_frames_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0, s, coords, n, B, status);
        hipLaunchKernelGGL(l0_frame_sort_kernel, dim3((unsigned)B), dim3(L0F_NT), 0, s, coords, n, g, slot16, (unsigned*)bl, bid, perm,
                           coords_raster, status);
        AL3D_CHECK_LAUNCH("l0_frame_sort_kernel");
        return AL3D_OK;
    }
    if (hipMemsetAsync(cnt, 0, L * 4, s) != hipSuccess) return al3d_fail(AL3D_ELAUNCH, "al3d_sp_raster_perm: memset failed");
    const unsigned nb = (unsigned)al3d_cdiv(n, 256);
    hipLaunchKernelGGL(l0_line_count_kernel, dim3(nb), dim3(256), 0, s, coords, n, g, cnt, slot);
    int rc = al3d_exclusive_scan_i32(cnt, base, L, scan_ws, s);
    if (rc != AL3D_OK) return rc;
    hipLaunchKernelGGL(l0_bucket_kernel, dim3(nb), dim3(256), 0, s, coords, n, g, base, slot, bx, bid);
    hipLaunchKernelGGL(l0_rank_kernel, dim3((unsigned)al3d_cdiv(L, 256)), dim3(256), 0, s, cnt, base, (int)L, g, bx, bid,
                       perm, coords_raster);
    AL3D_CHECK_LAUNCH("al3d_sp_raster_perm");
    return AL3D_OK;
}

// out[r] = rows[perm[r]] zero-padded from F to C channels (C % 8 == 0), f32 rows or pair rows
__global__ __launch_bounds__(256) void l0_gather_pad_kernel(const float* __restrict__ feat, const int* __restrict__ perm,
                                                            int64_t groups, int F, int C, int to_pair,
                                                            float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= groups) return;
    const int gpr = C / 8;
    const int64_t row = t / gpr;
    const int c0 = (int)(t - row * gpr) * 8;
    const int64_t src = perm ? perm[row] : row;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < F ? feat[src * F + c0 + e] : 0.f;
    float* dst = out + row * C + c0;
    if (to_pair) {
        uint4 hi, lo;
        sp_split8(v, hi, lo);
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 4) = lo;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

extern "C" int al3d_sp_rows_gather_pad_f32(const float* rows, const int* perm, int64_t n, int channels_in, int channels_out,
                                           int to_pair, float* out, void* stream)
{
    AL3D_REQUIRE(n >= 0 && channels_in >= 1 && channels_out >= channels_in && channels_out % 8 == 0,
                 "al3d_sp_rows_gather_pad_f32: channels_out must be a multiple of 8 and >= channels_in");
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(rows && out, "al3d_sp_rows_gather_pad_f32: null pointer");
    const int64_t groups = n * (channels_out / 8);
    hipLaunchKernelGGL(l0_gather_pad_kernel, dim3((unsigned)al3d_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream, rows,
                       perm, groups, channels_in, channels_out, to_pair, out);
    AL3D_CHECK_LAUNCH("l0_gather_pad_kernel");
    return AL3D_OK;
}

// =====================================================================================================
// 2. items: the live (tile, group) pairs of a tiled 27-tap table as a flat list
// =====================================================================================================
// item = int4 { lo, len | group << 16 | first-of-tile << 20 | last-of-tile << 21, tap mask of the tile, tile }
#define R16_FIRST (1 << 20)
#define R16_LAST (1 << 21)

__device__ __forceinline__ unsigned r16_groups_of(unsigned m27)
{
    unsigned g = 0u;
#pragma unroll
    for (int i = 0; i < 9; ++i) g |= ((m27 >> (3 * i)) & 7u) ? 1u << i : 0u;
    return g;
}

__global__ __launch_bounds__(256) void r16_count_kernel(const unsigned* __restrict__ tmask, int ntiles, int* __restrict__ cnt)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t > ntiles) return;
    cnt[t] = t < ntiles ? __popc(r16_groups_of(tmask[t])) : 0;
}

// one half-wave per tile: (lo, len) of each live group from the table's three kx columns, written at first[tile] + k
__global__ __launch_bounds__(256) void r16_items_kernel(const int* __restrict__ nbr, int64_t pitch, int n_out, int ntiles,
                                                        const unsigned* __restrict__ tmask, const int* __restrict__ first,
                                                        int4* __restrict__ items)
{
    const int tile = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r = threadIdx.x & 31;
    if (tile >= ntiles) return;                                              // half-wave uniform
    const unsigned tm = tmask[tile];
    const unsigned gm = r16_groups_of(tm);
    const int row = tile * 32 + r;
    int k = first[tile];
    const int nlive = __popc(gm);
    int seen = 0;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        if (!(gm >> g & 1u)) continue;                                       // half-wave uniform
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
        ++seen;
        if (r == 0) {
            const int len = mx - mn + 1;
            const int meta = (len > 0xffff ? 0xffff : len) | (g << 16) | (seen == 1 ? R16_FIRST : 0) | (seen == nlive ? R16_LAST : 0);
            items[k] = make_int4(mn, meta, (int)tm, tile);
        }
        ++k;
    }
    if (tile == ntiles - 1 && r == 0) items[first[ntiles]] = make_int4(0, 0, 0, ntiles);     // the dummy item past the end
}

extern "C" int64_t al3d_sp_tile_items_workspace_bytes(int n_out)
{
    const int64_t nt = al3d_cdiv(n_out > 0 ? n_out : 1, 32) + 1;
    return al3d_align(nt * 4, 256) + al3d_scan_workspace_bytes(nt);
}

// first: [ntiles + 1] ints (index of a tile's first item; first[ntiles] = number of items); items: room for
// 9 * ntiles + 1 int4s
extern "C" int al3d_sp_tile_items(const int* nbr, int64_t nbr_pitch, int K, int n_out, const unsigned* tile_mask,
                                  void* workspace, int* first, void* items, void* stream)
{
    AL3D_REQUIRE(K == 27 && n_out >= 0 && nbr_pitch >= n_out, "al3d_sp_tile_items: 27-tap tables only");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(nbr && tile_mask && workspace && first && items, "al3d_sp_tile_items: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)al3d_cdiv(n_out, 32);
    int* cnt = (int*)workspace;
    void* scan_ws = (char*)workspace + al3d_align((int64_t)(ntiles + 1) * 4, 256);
    hipLaunchKernelGGL(r16_count_kernel, dim3((unsigned)al3d_cdiv(ntiles + 1, 256)), dim3(256), 0, s, tile_mask, ntiles, cnt);
    int rc = al3d_exclusive_scan_i32(cnt, first, ntiles + 1, scan_ws, s);
    if (rc != AL3D_OK) return rc;
    hipLaunchKernelGGL(r16_items_kernel, dim3((unsigned)al3d_cdiv(ntiles, 8)), dim3(256), 0, s, nbr, nbr_pitch, n_out, ntiles,
                       tile_mask, first, (int4*)items);
    AL3D_CHECK_LAUNCH("al3d_sp_tile_items");
    return AL3D_OK;
}

// =====================================================================================================
// 3. weights: planes [2][Cout][27][16] f16 -> LDS image [27][2 planes][2 k-halves][Cout][8] f16
// =====================================================================================================
__global__ void r16_pack_kernel(const unsigned short* __restrict__ planes, int cout, unsigned short* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                      // one thread per 8-element piece
    const int total = 27 * 2 * 2 * cout;
    if (t >= total) return;
    const int n = t % cout, kb = (t / cout) % 2, pl = (t / (2 * cout)) % 2, tap = t / (4 * cout);
    const uint4 v = *reinterpret_cast<const uint4*>(planes + (((int64_t)pl * cout + n) * 27 + tap) * 16 + 8 * kb);
    *reinterpret_cast<uint4*>(out + (int64_t)t * 8) = v;
}

extern "C" int64_t al3d_sp_pack_r16_f16x3_elems(int cout) { return (cout == 16 || cout == 32) ? (int64_t)27 * 2 * 2 * cout * 8 : -1; }

extern "C" int al3d_sp_pack_r16_f16x3(const void* planes_f16x2, int cout, void* out_image, void* stream)
{
    AL3D_REQUIRE(cout == 16 || cout == 32, "al3d_sp_pack_r16_f16x3: Cout must be 16 or 32 (Cin = 16, 27 taps)");
    AL3D_REQUIRE(planes_f16x2 && out_image, "al3d_sp_pack_r16_f16x3: null pointer");
    const int total = 27 * 2 * 2 * cout;
    hipLaunchKernelGGL(r16_pack_kernel, dim3((unsigned)al3d_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)planes_f16x2, cout, (unsigned short*)out_image);
    AL3D_CHECK_LAUNCH("r16_pack_kernel");
    return AL3D_OK;
}

// =====================================================================================================
// 4. the convolution
// =====================================================================================================
// one tap's fragments (the per-row fallback)
template <int PLB>
__device__ __forceinline__ void r16_read_tap(gl_f32x4& alo, gl_f32x4& ahi, f16x8& wh, f16x8& wl, unsigned a0, unsigned a1,
                                             unsigned wb)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:%7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(alo), "=&v"(ahi), "=&v"(wh), "=&v"(wl) : "v"(a0), "v"(a1), "v"(wb), "n"(PLB) : "memory");
}
// eight accumulator registers -> the wave's transposition scratch (row offsets as immediates)
template <int RB>
__device__ __forceinline__ void r16_scr_write8(unsigned addr, float v0, float v1, float v2, float v3, float v4, float v5,
                                               float v6, float v7, int)
{
    // rows (e & 3) + 8 * (e >> 2) for e = 0..7: 0 1 2 3 8 9 10 11
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%9\n\tds_write_b32 %0, %3 offset:%10\n\t"
                 "ds_write_b32 %0, %4 offset:%11\n\tds_write_b32 %0, %5 offset:%12\n\tds_write_b32 %0, %6 offset:%13\n\t"
                 "ds_write_b32 %0, %7 offset:%14\n\tds_write_b32 %0, %8 offset:%15"
                 :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7),
                    "n"(1 * RB), "n"(2 * RB), "n"(3 * RB), "n"(8 * RB), "n"(9 * RB), "n"(10 * RB), "n"(11 * RB)
                 : "memory");
}
__device__ __forceinline__ void r16_lds_read16(gl_f32x4& d, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
}
__device__ __forceinline__ void r16_lds_read16x2(gl_f32x4& d0, gl_f32x4& d1, unsigned a0, unsigned a1)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d0), "=&v"(d1) : "v"(a0), "v"(a1) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Staging.  A first form of this kernel staged the ranges by LDS-DMA (global_load_lds) into a ring of LDS slots, with the
// indices and the residual tile DMA'd too.  Measured: the DMA path moves ~30 B/clk per CU whatever the
// locality -- with every request redirected to one zero line and no products the loop still took 60 % of the kernel's
// time -- and an item needed five DMA instructions.  Plain loads take the vector L1's 64 B/clk path; the rows wait in
// registers (12 VGPRs per item in flight), the indices and the residual pieces arrive in registers too: no LDS-DMA, no
// index ring, no residual ring, 5.5 KB of LDS per wave.  All vector memory operations are ordinary loads / stores, so
// hipcc's own counted vmcnt waits are exact as long as every pipeline step is unconditional; the long-range fallback keeps
// its loads inside asm blocks with their waits (a VMEM operation under a branch would make hipcc drain the ring).
template <int COUT, int NW, int P, int CAP, bool RES>
struct R16Cfg {
    static constexpr int PLB = 2 * COUT * 16, TAPB = 2 * PLB, W_BYTES = 27 * TAPB;
    static constexpr int NPC = CAP / 16;
    static constexpr int SLOT_BYTES = CAP * 64;
    static constexpr int EP_PITCH = COUT + 4;
    static constexpr int SCR_BYTES = 32 * EP_PITCH * 4;
    static constexpr int WAVE_BYTES = SLOT_BYTES + SCR_BYTES;
    static constexpr int ZERO_OFF = W_BYTES + NW * WAVE_BYTES;
    static constexpr int SMEM_BYTES = ZERO_OFF + 64;
    static_assert(COUT == 16 || COUT == 32, "output channels");
    static_assert(CAP % 16 == 0 && CAP >= 32 && P >= 2, "shape");
    static_assert(2 * TAPB + PLB < 65536, "ds_read immediate offsets");
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
    static_assert(!RES || COUT == 16, "residual form at 16 output channels only");
};

__device__ __forceinline__ void r16_load16_now(gl_i32x4& d, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void r16_lds_write16(unsigned addr, const gl_i32x4& v)
{
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
// A fragments of three taps (all lanes) / B fragments of three taps (the caller masks the lanes of columns >= COUT)
__device__ __forceinline__ void r16_read_a3(gl_f32x4 (&alo)[3], gl_f32x4 (&ahi)[3], const unsigned (&a0)[3], const unsigned (&a1)[3])
{
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
                 "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(alo[0]), "=&v"(ahi[0]), "=&v"(alo[1]), "=&v"(ahi[1]), "=&v"(alo[2]), "=&v"(ahi[2])
                 : "v"(a0[0]), "v"(a1[0]), "v"(a0[1]), "v"(a1[1]), "v"(a0[2]), "v"(a1[2]) : "memory");
}
template <int TAPB, int PLB>
__device__ __forceinline__ void r16_read_b3(f16x8 (&wh)[3], f16x8 (&wl)[3], unsigned wb)
{
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:%7\n\tds_read_b128 %2, %6 offset:%8\n\t"
                 "ds_read_b128 %3, %6 offset:%9\n\tds_read_b128 %4, %6 offset:%10\n\tds_read_b128 %5, %6 offset:%11"
                 : "+v"(wh[0]), "+v"(wl[0]), "+v"(wh[1]), "+v"(wl[1]), "+v"(wh[2]), "+v"(wl[2])
                 : "v"(wb), "n"(PLB), "n"(TAPB), "n"(TAPB + PLB), "n"(2 * TAPB), "n"(2 * TAPB + PLB) : "memory");
}

template <int PLB>
__device__ __forceinline__ void r16_read_b1(f16x8& wh, f16x8& wl, unsigned wb)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "+v"(wh), "+v"(wl) : "v"(wb), "n"(PLB) : "memory");
}

template <int COUT, int NW, int P, int CAP, bool RES>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 3 : 1)) void sp_conv_r16_kernel(const float* __restrict__ fin, const int* __restrict__ nbr,
                                                                int pitch, const int4* __restrict__ items,
                                                                const int* __restrict__ first, int ntiles, int tpw,
                                                                const unsigned char* __restrict__ wimg,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ residual, int relu,
                                                                float* __restrict__ fout, int n_out, int io, int abl)
{
    using C = R16Cfg<COUT, NW, P, CAP, RES>;
    constexpr int NPC = C::NPC, TAPB = C::TAPB, PLB = C::PLB, EP_PITCH = C::EP_PITCH;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    for (int o = tid * 16; o < C::W_BYTES; o += 64 * NW * 16)
        *reinterpret_cast<uint4*>(smem + o) = *reinterpret_cast<const uint4*>(wimg + o);
    if (tid < 16) reinterpret_cast<float*>(smem + C::ZERO_OFF)[tid] = 0.f;
    const bool pair_ep = (io & (SP_IO_OUT_PAIR | SP_IO_RES_PAIR)) != 0;       // uniform
    constexpr int Q4 = COUT / 4, Q8 = COUT / 8;
    const int ec = pair_ep ? (lane % Q8) * 8 : (lane % Q4) * 4;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool use = pair_ep || e < 4;
        sc[e] = use ? scale[ec + e] : 1.f;
        sh[e] = use ? (shift ? shift[ec + e] : 0.f) : 0.f;
    }
    __syncthreads();

    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tile0 = (wg * NW + wave) * tpw;
    if (tile0 >= ntiles) return;                                              // wave-uniform, after the barrier
    const int tile1 = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
    const int j0 = first[tile0], j1 = first[tile1], jd = first[ntiles];

    const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;
    const unsigned sA = smem_base + C::W_BYTES + wave * C::WAVE_BYTES;
    const unsigned scr_base = sA + C::SLOT_BYTES;
    const unsigned zero_base = smem_base + C::ZERO_OFF;
    const int jd4 = lane >> 2;
    const int chk = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const unsigned wb_lane = smem_base + fh * (COUT * 16) + (fr % COUT) * 16;
    const unsigned wr_lane = sA + lane * 16;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // the ring: indices, rows and (residual form) the residual pieces of the items in flight
    int xi[P][3];
    gl_i32x4 rw[P][NPC];
    gl_i32x4 rr[P][2];

    auto issue = [&](const int4& d, int s, bool real) __attribute__((always_inline)) {
        const int lo = d.x, len = d.y & 0xffff, g = (d.y >> 16) & 15, tile = d.w;
        // every load is unconditional (redirected sources for dummy / irregular items): hipcc keeps counted waits
        const bool xr = real && !(abl & 2);
        const int* xp = xr ? nbr + (int64_t)(3 * g) * pitch + (int64_t)tile * 32 + fr : g_glds_neg1 + fr;
        const int64_t xs = xr ? pitch : 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) xi[s][kx] = xp[kx * xs];
        const bool regular = real && len > 0 && len <= CAP && !(abl & 1);     // wave-uniform
        const char* base = regular ? reinterpret_cast<const char*>(fin) + (int64_t)lo * 64 : reinterpret_cast<const char*>(g_glds_zero);
        const int stride = regular ? 64 : 0;
        const int last = regular ? len - 1 : 0;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            int r = 16 * i + jd4;
            r = r < last ? r : last;
            rw[s][i] = *reinterpret_cast<const gl_i32x4*>(base + r * stride + chk);
        }
        if constexpr (RES) {
            // the residual pieces of the tile in the epilogue's own lane order, with the tile's LAST item
            const bool rl_ = real && (d.y & R16_LAST) && !(abl & 32);
            const char* rb = rl_ ? reinterpret_cast<const char*>(residual) : reinterpret_cast<const char*>(g_glds_zero);
            if (!pair_ep) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int row = tile * 32 + (lane + 64 * i) / Q4;
                    row = row < n_out ? row : n_out - 1;
                    rr[s][i] = *reinterpret_cast<const gl_i32x4*>(rb + (rl_ ? ((int64_t)row * COUT + ec) * 4 : 0));
                }
            } else {
                int row = tile * 32 + lane / Q8;
                row = row < n_out ? row : n_out - 1;
                const char* p = rb + (rl_ ? ((int64_t)row * COUT + ec) * 4 : 0);
                rr[s][0] = *reinterpret_cast<const gl_i32x4*>(p);
                rr[s][1] = *reinterpret_cast<const gl_i32x4*>(p + (rl_ ? 16 : 0));
            }
        }
    };

    auto mac = [&](const gl_f32x4& vlo, const gl_f32x4& vhi, const f16x8& wh, const f16x8& wl) __attribute__((always_inline)) {
        f16x8 ah, al;
        if (io & SP_IO_IN_PAIR) {
            ah = __builtin_bit_cast(f16x8, vlo);
            al = __builtin_bit_cast(f16x8, vhi);
        } else {
            gl_split8_f16(vlo, vhi, ah, al);
        }
        const f16x8 wd = gl_lift_down(wh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wd, acc, 0, 0, 0);       // smallest first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
    };

    constexpr int EPV = COUT == 16 ? 8 : 16;
    float eo[EPV];
    auto ep_read = [&](int s) __attribute__((always_inline)) {
        const unsigned wa = scr_base + ((4 * fh) * EP_PITCH + fr) * 4;
        if (fr < COUT) {
            r16_scr_write8<EP_PITCH * 4>(wa, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], 0);
            r16_scr_write8<EP_PITCH * 4>(wa + 16 * EP_PITCH * 4, acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15], 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!pair_ep) {
#pragma unroll
            for (int i = 0; i < (32 * Q4) / 64; ++i) {
                const int idx = lane + 64 * i;
                const int rl = idx / Q4, c4 = (idx % Q4) * 4;                 // c4 == ec
                gl_f32x4 v;
                r16_lds_read16(v, scr_base + (rl * EP_PITCH + c4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = v[e] * sc[e] + sh[e];
                    if constexpr (RES) o += __builtin_bit_cast(gl_f32x4, rr[s][i < 2 ? i : 0])[e];
                    if (relu) o = o <= 0.f ? 0.f : o;                           // NaN propagates, like torch.relu
                    eo[4 * i + e] = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < (32 * Q8) / 64; ++i) {
                const int grp = lane + 64 * i;
                const int rl = grp / Q8, c8 = (grp % Q8) * 8;                 // c8 == ec
                gl_f32x4 a, b;
                r16_lds_read16x2(a, b, scr_base + (rl * EP_PITCH + c8) * 4, scr_base + (rl * EP_PITCH + c8 + 4) * 4);
                float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] + sh[e];
                if constexpr (RES) {
                    float r[8];
                    if (io & SP_IO_RES_PAIR) {
                        sp_unsplit8(__builtin_bit_cast(uint4, rr[s][0]), __builtin_bit_cast(uint4, rr[s][1]), r);
                    } else {
                        const gl_f32x4 ra = __builtin_bit_cast(gl_f32x4, rr[s][0]), rb = __builtin_bit_cast(gl_f32x4, rr[s][1]);
                        r[0] = ra[0]; r[1] = ra[1]; r[2] = ra[2]; r[3] = ra[3]; r[4] = rb[0]; r[5] = rb[1]; r[6] = rb[2]; r[7] = rb[3];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] <= 0.f ? 0.f : v[e];
                }
                if (io & SP_IO_OUT_PAIR) {
                    uint4 hi, lo;
                    sp_split8(v, hi, lo);
                    const gl_f32x4 fh4 = __builtin_bit_cast(gl_f32x4, hi), fl4 = __builtin_bit_cast(gl_f32x4, lo);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { eo[8 * i + e] = fh4[e]; eo[8 * i + 4 + e] = fl4[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) eo[8 * i + e] = v[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    };
    auto ep_store = [&](int tile) __attribute__((always_inline)) {
        const int wrow0 = tile * 32;
        if (!pair_ep) {
#pragma unroll
            for (int i = 0; i < (32 * Q4) / 64; ++i) {
                const int row = wrow0 + (lane + 64 * i) / Q4;
                if (row < n_out)
                    *reinterpret_cast<float4*>(fout + (int64_t)row * COUT + ec) = make_float4(eo[4 * i], eo[4 * i + 1], eo[4 * i + 2], eo[4 * i + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < (32 * Q8) / 64; ++i) {
                const int row = wrow0 + (lane + 64 * i) / Q8;
                if (row < n_out) {
                    float* dst = fout + (int64_t)row * COUT + ec;
                    *reinterpret_cast<float4*>(dst) = make_float4(eo[8 * i], eo[8 * i + 1], eo[8 * i + 2], eo[8 * i + 3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(eo[8 * i + 4], eo[8 * i + 5], eo[8 * i + 6], eo[8 * i + 7]);
                }
            }
        }
    };

    f16x8 bwh = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, bwl = bwh;                         // B fragments: lanes of columns >= COUT keep stale values

    auto consume = [&](const int4& d, int s) __attribute__((always_inline)) {
        const int lo = d.x, len = d.y & 0xffff, g = (d.y >> 16) & 15;
        const unsigned tm = (unsigned)d.z >> (3 * g);                          // the group's three tap bits
        const unsigned wb = wb_lane + (3 * g) * TAPB;
        if (len <= CAP) {
            // the staged rows: registers -> the wave's slot (the previous item's fragment reads have completed)
#pragma unroll
            for (int i = 0; i < NPC; ++i) r16_lds_write16(wr_lane + i * 1024, rw[s][i]);
            unsigned a0[3], a1[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int id = xi[s][kx];
                const bool ok = id >= 0;
                const int local = id - lo;
                const unsigned f = (unsigned)(local >> 2) & 3u;
                a0[kx] = ok ? sA + (unsigned)local * 64u + (((2u * fh) ^ f) << 4) : zero_base;
                a1[kx] = ok ? a0[kx] ^ 16u : zero_base + 16u;
            }
            // one tap at a time (16 fragment registers instead of 48: the other waves of the SIMD cover the LDS round trips);
            // dead taps are neither read nor multiplied
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (!((tm >> kx & 1u) && !(abl & 4))) continue;                 // wave-uniform
                gl_f32x4 vlo, vhi;
                if (!(abl & 8)) {
                    if (fr < COUT) r16_read_b1<PLB>(bwh, bwl, wb + kx * TAPB);   // columns >= COUT are never stored
                    gl_lds_read_a(vlo, vhi, a0[kx], a1[kx]);                    // ... and this wait covers both
                } else {
                    vlo = vhi = gl_f32x4{0.f, 0.f, 0.f, 0.f};
                }
                mac(vlo, vhi, bwh, bwl);
            }
        } else {
            // long range: per-row gather of each tap into rows 0..31 of the slot; loads + waits inside asm blocks
            gl_static_for<3>([&](auto KX) {
                constexpr int kx = decltype(KX)::value;
                const int id = xi[s][kx];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idr = __shfl(id, 16 * i + jd4);
                    const char* src = idr >= 0 ? reinterpret_cast<const char*>(fin) + (int64_t)idr * 64 + chk
                                               : reinterpret_cast<const char*>(g_glds_zero) + (lane & 3) * 16;
                    gl_i32x4 v;
                    r16_load16_now(v, src);
                    r16_lds_write16(wr_lane + i * 1024, v);
                }
                const bool ok = id >= 0;
                const unsigned f = (unsigned)(fr >> 2) & 3u;
                const unsigned a0 = ok ? sA + (unsigned)fr * 64u + (((2u * fh) ^ f) << 4) : zero_base;
                const unsigned a1 = ok ? a0 ^ 16u : zero_base + 16u;
                gl_f32x4 vlo, vhi;
                f16x8 bh, bl;
                r16_read_tap<PLB>(vlo, vhi, bh, bl, a0, a1, wb + kx * TAPB);
                if (tm >> kx & 1u) mac(vlo, vhi, bh, bl);
            });
        }
        if (d.y & R16_LAST) ep_read(s);
    };

    int4 q[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const int j = j0 + s;
        q[s] = items[j < j1 ? j : jd];
        issue(q[s], s, j < j1);
    }
    for (int jb = j0; jb < j1; jb += P) {
        gl_static_for<P>([&](auto S) {
            constexpr int s = decltype(S)::value;
            // every step runs (a step past the end consumes the dummy item: no product, no epilogue): a conditional step
            // would leave hipcc a path on which the other slot's loads were never issued, and it then waits for this
            // slot's loads with vmcnt(0) -- no prefetch distance left
            const int jn = jb + s + P;
            const int4 nxt = items[jn < j1 ? jn : jd];
            consume(q[s], s);
            const bool tile_done = (q[s].y & R16_LAST) != 0;
            const int tile_id = q[s].w;
            q[s] = nxt;
            issue(q[s], s, jn < j1);
            if (tile_done && !(abl & 16)) ep_store(tile_id);                   // wave-uniform
        });
    }
}

#define R16_LAUNCH(CO, NW, P, CAP, RES)                                                                            \
    do {                                                                                                             \
        const int per_wg = (NW) * tpw;                                                                               \
        hipLaunchKernelGGL((sp_conv_r16_kernel<CO, NW, P, CAP, RES>), dim3((unsigned)al3d_cdiv(ntiles, per_wg)),     \
                           dim3(64 * (NW)), 0, s, fin, nbr, nbr_pitch, (const int4*)items, first, ntiles, tpw,         \
                           (const unsigned char*)wgt_image, scale, shift, residual, relu, fout, n_out, io, abl);      \
        AL3D_CHECK_LAUNCH("sp_conv_r16_kernel");                                                                   \
        return AL3D_OK;                                                                                              \
    } while (0)

// Same contract as al3d_sp_conv_rng_f16x3 for Cin = 16: a tiled 27-tap table (submanifold or strided) + its item list
// (al3d_sp_tile_items), weight image of al3d_sp_pack_r16_f16x3.  Rows of the INPUT level should be in raster order for the
// ranges to be short (any order is correct: long ranges take the per-row path).  tiles_per_wave <= 0: default.
extern "C" int al3d_sp_conv_r16_f16x3(const float* fin, const int* nbr, int nbr_pitch, const void* items, const int* first,
                                      int K, const void* wgt_image, int cin, int cout, const float* scale, const float* shift,
                                      const float* residual, int relu, float* fout, int n_out, int io, int tiles_per_wave,
                                      void* stream)
{
    AL3D_REQUIRE(K == 27 && cin == 16 && n_out >= 0, "al3d_sp_conv_r16_f16x3: 27-tap layers with 16 input channels only");
    AL3D_REQUIRE(io >= 0 && io < 8, "al3d_sp_conv_r16_f16x3: bad io flags");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && items && first && wgt_image && fout, "al3d_sp_conv_r16_f16x3: null pointer");
    AL3D_REQUIRE(scale, "al3d_sp_conv_r16_f16x3: scale carries the weight exponent and is required");
    AL3D_REQUIRE(nbr_pitch >= n_out && nbr_pitch % 256 == 0, "al3d_sp_conv_r16_f16x3: nbr_pitch must be al3d_sp_table_pitch(n_out)");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)al3d_cdiv(n_out, 32);
    // tiles per wave: longer item streams amortise the pipeline's fill (8: -1.5 % at 240k tiles), shorter ones keep small
    // launches balanced over the CUs (4: -7 % at 60k tiles); AL3D_R16_TPW (dev knob) overrides: 4 / 8 / 16 / 32 measure
    // 2,207 / 2,211 / 2,215 / 2,204 frames/s in the bench with the 4-wave workgroups
    static const int tpw_env = getenv("AL3D_R16_TPW") ? atoi(getenv("AL3D_R16_TPW")) : 0;
    const int tpw = tiles_per_wave > 0 ? tiles_per_wave : tpw_env > 0 ? tpw_env : (ntiles >= 150000 ? 8 : 4);
    // workgroup shapes: (waves, items in flight per wave); AL3D_R16_SHAPE (dev knob).  The kernel is bound by the latency chain
    // of a wave's item times the resident waves: 12 waves per CU at two items in flight (8 x 3 measured +4..18 %).  At 16
    // output channels those twelve waves are THREE workgroups of four (default, shape 4) rather than one of twelve (shape 0):
    // a 12-wave workgroup needs a whole CU's registers at once, so every CU on which a side-stream workgroup sits is closed to
    // it until that drains -- the first level-0 layer of a batch ran 2.0 ms beside the previous batch's decode + NMS against
    // 1.0 ms now (serial: 625 -> 587 us per layer; bench +0.8 %, tools/ab_r16_shape.sh).
    static const int shape = getenv("AL3D_R16_SHAPE") ? atoi(getenv("AL3D_R16_SHAPE")) : 4;
#ifdef AL3D_R16_ABLATE
    // tuning build only (make EXTRA=-DAL3D_R16_ABLATE, tools/ablate_l0.sh): runtime ablations that drop rows / indices /
    // products / stores -- never reachable from a stray environment variable in the shipped library
    static const int abl = getenv("AL3D_R16_ABL") ? atoi(getenv("AL3D_R16_ABL")) : 0;
#else
    constexpr int abl = 0;
#endif
    {
        if (cout == 16) {
            if (residual) {
                if (shape == 2) R16_LAUNCH(16, 8, 3, 48, true);
                if (shape == 0) R16_LAUNCH(16, 12, 2, 48, true);
                R16_LAUNCH(16, 4, 2, 48, true);
            }
            if (shape == 2) R16_LAUNCH(16, 8, 3, 48, false);
            if (shape == 0) R16_LAUNCH(16, 12, 2, 48, false);
            R16_LAUNCH(16, 4, 2, 48, false);
        }
        if (cout == 32) {
            AL3D_REQUIRE(!residual, "al3d_sp_conv_r16_f16x3: no residual form at 32 output channels");
            if (shape == 2) R16_LAUNCH(32, 8, 3, 64, false);
            R16_LAUNCH(32, 10, 2, 64, false);
        }
    }
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_r16_f16x3: no kernel for Cout=%d", cout);
}
