// 3-D sparse convolution middle encoder (submanifold + strided), gfx950.
//
// Semantics follow spconv (call sites det3d/models/backbones/scn.py:28-97,316-392; the
// in-tree statement of the rulebook is the vendored spconv-1.0 under
// bevfusion/mmdet3d/ops/spconv/include/spconv/geometry.h:25-82,145-194,248-298 and
// spconv_ops.h:260-361):
//     out[o] = sum_k  W[k]^T . in[o*stride - pad + k]          (k over kz,ky,kx, x fastest)
//   * SubMConv3d: output sites == input sites (pad = k/2, stride 1)
//   * SparseConv3d: output sites = every in-bounds o reached by at least one input
// MI355X design: instead of spconv's per-offset gather -> GEMM -> scatter-add (27
// launches, atomics, run-to-run summation order), the rulebook is stored output-major
// (nbr[k][o] = input row or -1, tap-major, built from a dense per-level index grid that lives in
// HBM) and ONE kernel per layer walks k for a tile of output rows: gathered input rows are
// staged in LDS, products accumulate in registers in a fixed (k, ci) order, and the
// BN(eval)/bias/residual/ReLU epilogue is fused, so every activation row makes one HBM
// round trip and results are deterministic.
#include "al3d_common.h"
#include "al3d_scan.h"

struct SpDims { int B, D, H, W; };

__device__ __forceinline__ int64_t sp_cell(const SpDims& g, int b, int z, int y, int x)
{
    return (((int64_t)b * g.D + z) * g.H + y) * g.W + x;
}

// grid[cell(coords[i])] = (mode ? i : -1)
__global__ void sp_scatter_index_kernel(const int* __restrict__ coords, int n, SpDims g,
                                        int* __restrict__ grid, int mode)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
    grid[sp_cell(g, c.x, c.y, c.z, c.w)] = mode ? i : -1;
}

// Output-major rulebook for a submanifold conv: nbr[i][k] = row at coords[i] + (k - k/2).
// One thread per (output row, kz, ky): its kw lookups are adjacent grid cells (x-1, x, x+1), i.e. one
// memory transaction instead of kw, and the row's coordinates are read once per tap row.
__global__ void sp_subm_table_kernel(const int* __restrict__ coords, int n, SpDims g,
                                     const int* __restrict__ grid, int kd, int kh, int kw,
                                     int* __restrict__ nbr)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * kd * kh) return;
    const int kzy = (int)(e / n), i = (int)(e % n);     // tap-major: coalesced over i
    const int ky = kzy % kh, kz = kzy / kh;
    const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
    const int z = c.y + kz - kd / 2, y = c.z + ky - kh / 2, x0 = c.w - kw / 2;
    const bool row_ok = z >= 0 && z < g.D && y >= 0 && y < g.H;
    const int64_t base = row_ok ? sp_cell(g, c.x, z, y, 0) : 0;
    for (int kx = 0; kx < kw; ++kx) {
        const int x = x0 + kx;
        int v = -1;
        if (row_ok && x >= 0 && x < g.W) v = grid[base + x];
        nbr[((int64_t)kzy * kw + kx) * n + i] = v;
    }
}

struct SpConvGeom { int kd, kh, kw, sd, sh, sw, pd, ph, pw; };

// Strided conv, step 1: every input site claims the output sites it feeds; the first claimer
// appends the site to coords_out (row order is arbitrary; nothing downstream depends on it).
// One thread per input: along each axis only the offsets k = (c + pad) mod stride (+ stride..)
// land on an integer output coordinate, so the loop visits exactly the valid pairs.
__global__ void sp_down_claim_kernel(const int* __restrict__ coords_in, int n_in, SpConvGeom q,
                                     SpDims go, int* __restrict__ grid_out,
                                     int* __restrict__ coords_out, int* __restrict__ counter, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const int* c = coords_in + 4 * i;
    const int b = c[0], az = c[1] + q.pd, ay = c[2] + q.ph, ax = c[3] + q.pw;
    for (int kz = az % q.sd; kz < q.kd; kz += q.sd) {
        const int oz = (az - kz) / q.sd;
        if (az - kz < 0 || oz >= go.D) continue;
        for (int ky = ay % q.sh; ky < q.kh; ky += q.sh) {
            const int oy = (ay - ky) / q.sh;
            if (ay - ky < 0 || oy >= go.H) continue;
            for (int kx = ax % q.sw; kx < q.kw; kx += q.sw) {
                const int ox = (ax - kx) / q.sw;
                if (ax - kx < 0 || ox >= go.W) continue;
                int* cell = grid_out + sp_cell(go, b, oz, oy, ox);
                // cheap read first: most sites are already claimed by a neighbouring input
                if (__hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) continue;
                if (atomicCAS(cell, -1, -2) == -1) {
                    const int row = atomicAdd(counter, 1);
                    if (row < cap) {
                        coords_out[4 * row + 0] = b; coords_out[4 * row + 1] = oz;
                        coords_out[4 * row + 2] = oy; coords_out[4 * row + 3] = ox;
                    }
                    atomicExch(cell, row);      // the site's row id, read by the next level's rulebook
                }
            }
        }
    }
}

// Deterministic alternative to the claim kernel: (1) every input marks the output sites it
// feeds (plain idempotent stores), (2) an exclusive scan over the marked cells numbers the sites
// in raster order (b, z, y, x), (3) the cells receive their row id and coords_out is written.
// Raster order makes consecutive rows spatial neighbours, which the conv kernels exploit when
// they skip kernel offsets that are empty for a whole 32-row tile.
// Output sites of a strided conv in raster order.  The dense output grid is touched as one BYTE per
// cell by the marking pass (plain idempotent stores; device-scope atomicOr on a bit mask was measured
// 10x slower -- the per-XCD L2s are not coherent, so such atomics execute at the memory side), then
// packed to a bit mask + popcount per 32-cell word; the popcounts are scanned and one thread per
// non-empty word numbers its set bits in ascending order -- the same (b, z, y, x) raster numbering a
// cell-wise scan gives, for 1/16 of its traffic.
// Blocked numbering (al3d_sp_down_sites_blocked): the flag map is laid out block-major -- 8 x 8 (y, x) columns
// over all z, cell' = ((((b NYB + y/8) NXB + x/8) D + z) 8 + y%8) 8 + x%8 -- so the same pack / scan / assign passes
// number the sites column by column: consecutive rows form compact patches whose z neighbours are in the patch
// (csrc/spconv_blk.hip).  grid_out keeps the plain (b, z, y, x) layout.
__device__ __forceinline__ int64_t sp_cell_blocked(const SpDims& g, int b, int z, int y, int x)
{
    const int nyb = (g.H + 7) >> 3, nxb = (g.W + 7) >> 3;
    return (((((int64_t)b * nyb + (y >> 3)) * nxb + (x >> 3)) * g.D + z) << 6) | ((y & 7) << 3) | (x & 7);
}

template <bool BLOCKED>
__global__ void sp_down_mark_kernel(const int* __restrict__ coords_in, int n_in, SpConvGeom q, SpDims go,
                                    unsigned char* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const int* c = coords_in + 4 * i;
    const int b = c[0], az = c[1] + q.pd, ay = c[2] + q.ph, ax = c[3] + q.pw;
    for (int kz = az % q.sd; kz < q.kd; kz += q.sd) {
        const int oz = (az - kz) / q.sd;
        if (az - kz < 0 || oz >= go.D) continue;
        for (int ky = ay % q.sh; ky < q.kh; ky += q.sh) {
            const int oy = (ay - ky) / q.sh;
            if (ay - ky < 0 || oy >= go.H) continue;
            for (int kx = ax % q.sw; kx < q.kw; kx += q.sw) {
                const int ox = (ax - kx) / q.sw;
                if (ax - kx < 0 || ox >= go.W) continue;
                flags[BLOCKED ? sp_cell_blocked(go, b, oz, oy, ox) : sp_cell(go, b, oz, oy, ox)] = 1;
            }
        }
    }
}

// 32 flag bytes -> one mask word + its popcount (flags is padded to a multiple of 32 bytes)
__global__ void sp_down_pack_kernel(const unsigned char* __restrict__ flags, int64_t words,
                                    unsigned* __restrict__ bits, int* __restrict__ cnt)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint4 a = reinterpret_cast<const uint4*>(flags)[2 * w], b = reinterpret_cast<const uint4*>(flags)[2 * w + 1];
    const unsigned v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // bytes are 0 or 1: gather bit 0 of each of the four bytes
        const unsigned x = v[i];
        m |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * i);
    }
    bits[w] = m;
    cnt[w] = __popc(m);
}

template <bool BLOCKED>
__global__ void sp_down_assign_kernel(const unsigned* __restrict__ bits, const int* __restrict__ wscan,
                                      int64_t words, SpDims go, int* __restrict__ grid_out,
                                      int* __restrict__ coords_out, int cap)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    unsigned m = bits[w];
    if (!m) return;
    int row = wscan[w];
    while (m) {
        const int bit = __builtin_ctz(m);
        m &= m - 1u;
        const int64_t cell = w * 32 + bit;
        int64_t t = cell;
        int x, y, z;
        if constexpr (BLOCKED) {
            const int nyb = (go.H + 7) >> 3, nxb = (go.W + 7) >> 3;
            const int xx = (int)(t & 7), yy = (int)((t >> 3) & 7);
            t >>= 6;
            z = (int)(t % go.D); t /= go.D;
            x = (int)(t % nxb) * 8 + xx; t /= nxb;
            y = (int)(t % nyb) * 8 + yy; t /= nyb;
            grid_out[sp_cell(go, (int)t, z, y, x)] = row;
        } else {
            x = (int)(t % go.W); t /= go.W;
            y = (int)(t % go.H); t /= go.H;
            z = (int)(t % go.D); t /= go.D;
            grid_out[cell] = row;
        }
        if (row < cap) *reinterpret_cast<int4*>(coords_out + 4 * (int64_t)row) = make_int4((int)t, z, y, x);
        ++row;
    }
}

__global__ void sp_count_tail_kernel(const unsigned* __restrict__ bits, const int* __restrict__ wscan,
                                     int64_t words, int* __restrict__ counter)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *counter = wscan[words - 1] + __popc(bits[words - 1]);
}

// Strided conv, step 2: nbr[k][o] = input row at o*stride - pad + k (grid_in lookup); one thread per
// (output row, kz, ky) -- its kw lookups are adjacent input cells.
__global__ void sp_down_table_kernel(const int* __restrict__ coords_out, int n_out, SpConvGeom q,
                                     SpDims gi, const int* __restrict__ grid_in, int* __restrict__ nbr)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n_out * q.kd * q.kh) return;
    const int kzy = (int)(e / n_out), o = (int)(e % n_out);   // tap-major
    const int ky = kzy % q.kh, kz = kzy / q.kh;
    const int4 c = *reinterpret_cast<const int4*>(coords_out + 4 * (int64_t)o);
    const int z = c.y * q.sd - q.pd + kz, y = c.z * q.sh - q.ph + ky, x0 = c.w * q.sw - q.pw;
    const bool row_ok = z >= 0 && z < gi.D && y >= 0 && y < gi.H;
    const int64_t base = row_ok ? sp_cell(gi, c.x, z, y, 0) : 0;
    for (int kx = 0; kx < q.kw; ++kx) {
        const int x = x0 + kx;
        int v = -1;
        if (row_ok && x >= 0 && x < gi.W) v = grid_in[base + x];
        nbr[((int64_t)kzy * q.kw + kx) * n_out + o] = v;
    }
}

// three adjacent cells of a grid row with ONE 12-byte load (dword-aligned): the texture addresser then handles one request
// per lane instead of three (the lookups of a wave are 64 different lines either way)
__device__ __forceinline__ void sp_probe3(const int* __restrict__ grid, int64_t base, int x0, int W, bool row_ok, int (&v)[3])
{
    v[0] = v[1] = v[2] = -1;
    if (!row_ok) return;
    if (x0 >= 0 && x0 + 2 < W) {
        typedef int sp_i32x3 __attribute__((ext_vector_type(3)));
        sp_i32x3 t;
        asm volatile("global_load_dwordx3 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(grid + base + x0) : "memory");
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2];
    } else {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x = x0 + kx;
            if (x >= 0 && x < W) v[kx] = grid[base + x];
        }
    }
}

// ------------------------------------------------------------------ tiled rulebook (csrc/spconv_glds.hip)
// The same tables with a row pitch that is a multiple of 64 (rows >= n hold -1), so that the 32 entries of
// one (tap, 32-row tile) are one aligned 128-byte line, plus tmask[tile] = the set of taps with at least
// one neighbour in that tile: the conv kernel reads one word per tile instead of scanning 27 x 32 entries.
__global__ void sp_subm_table_tiles_kernel(const int* __restrict__ coords, int n, int pitch, SpDims g,
                                           const int* __restrict__ grid, int kd, int kh, int kw,
                                           int* __restrict__ nbr, unsigned* __restrict__ tmask)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // exact grid: pitch * kd * kh threads
    const int kzy = (int)(e / pitch), i = (int)(e % pitch);
    const int ky = kzy % kh, kz = kzy / kh;
    int z = 0, y = 0, x0 = 0, b = 0;
    bool row_ok = false;
    if (i < n) {
        const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
        b = c.x; z = c.y + kz - kd / 2; y = c.z + ky - kh / 2; x0 = c.w - kw / 2;
        row_ok = z >= 0 && z < g.D && y >= 0 && y < g.H;
    }
    const int64_t base = row_ok ? sp_cell(g, b, z, y, 0) : 0;
    const int lane = threadIdx.x & 63;
    int v3[3];
    if (kw == 3) sp_probe3(grid, base, x0, g.W, row_ok, v3);
    for (int kx = 0; kx < kw; ++kx) {
        const int x = x0 + kx;
        int v = -1;
        if (kw == 3) v = v3[kx];
        else if (row_ok && x >= 0 && x < g.W) v = grid[base + x];
        const int k = kzy * kw + kx;
        nbr[(int64_t)k * pitch + i] = v;
        const unsigned long long bal = __ballot(v >= 0);                    // the wave's 64 rows = tiles i/32, i/32 + 1
        if (lane == 0 && (bal & 0xffffffffull)) atomicOr(&tmask[i >> 5], 1u << k);
        if (lane == 32 && (bal >> 32)) atomicOr(&tmask[i >> 5], 1u << k);
    }
}

// 3 x 3 x 3 tables, ONE thread per output row (both the submanifold and the strided form: input position of tap
// (kz, ky, kx) = o * stride - pad + k).  The nine (kz, ky) probes of a row are nine independent 12-byte loads issued
// back to back (one wait), its coordinates are read once instead of nine times, and the per-tile tap masks come from
// ballots kept in scalar registers and are stored with plain writes: no atomics, no memset of the mask array.  Loads
// are unconditional -- a probe outside the grid (or of a padding row) reads the three cells at a clamped position and
// is discarded -- so that hipcc keeps them together.  Same table, same masks as the (row, kz, ky)-per-thread kernels.
typedef int sp_i32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ int sp_pick3(const sp_i32x3& t, int idx)
{
    return idx <= 0 ? t[0] : (idx == 1 ? t[1] : t[2]);
}
__global__ __launch_bounds__(256) void sp_table_rows27_kernel(const int* __restrict__ coords, int n, int pitch, SpConvGeom q,
                                                              SpDims g, const int* __restrict__ grid,
                                                              int* __restrict__ nbr, unsigned* __restrict__ tmask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // exact grid: pitch threads
    const int lane = threadIdx.x & 63;
    int b = 0, z0 = 0, y0 = 0, x0 = 0;
    const bool live = i < n;
    if (live) {
        const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
        b = c.x; z0 = c.y * q.sd - q.pd; y0 = c.z * q.sh - q.ph; x0 = c.w * q.sw - q.pw;
    }
    const int xc = x0 < 0 ? 0 : (x0 > g.W - 3 ? g.W - 3 : x0);
    const int shift = x0 - xc;
    const int* addr[9];
    bool ok[9];
#pragma unroll
    for (int kzy = 0; kzy < 9; ++kzy) {
        const int z = z0 + kzy / 3, y = y0 + kzy % 3;
        ok[kzy] = live && z >= 0 && z < g.D && y >= 0 && y < g.H;
        addr[kzy] = grid + (ok[kzy] ? sp_cell(g, b, z, y, xc) : 0);
    }
    sp_i32x3 t[9];
    asm volatile("global_load_dwordx3 %0, %9, off\n\t"
                 "global_load_dwordx3 %1, %10, off\n\t"
                 "global_load_dwordx3 %2, %11, off\n\t"
                 "global_load_dwordx3 %3, %12, off\n\t"
                 "global_load_dwordx3 %4, %13, off\n\t"
                 "global_load_dwordx3 %5, %14, off\n\t"
                 "global_load_dwordx3 %6, %15, off\n\t"
                 "global_load_dwordx3 %7, %16, off\n\t"
                 "global_load_dwordx3 %8, %17, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                   "=&v"(t[8])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]),
                   "v"(addr[7]), "v"(addr[8])
                 : "memory");
    unsigned mlo = 0u, mhi = 0u;
#pragma unroll
    for (int kzy = 0; kzy < 9; ++kzy) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x = x0 + kx;
            const int v = (ok[kzy] && x >= 0 && x < g.W) ? sp_pick3(t[kzy], kx + shift) : -1;
            const int k = kzy * 3 + kx;
            nbr[(int64_t)k * pitch + i] = v;
            const unsigned long long bal = __ballot(v >= 0);
            mlo |= (bal & 0xffffffffull) ? 1u << k : 0u;
            mhi |= (bal >> 32) ? 1u << k : 0u;
        }
    }
    if (lane == 0) tmask[i >> 5] = mlo;
    if (lane == 32) tmask[i >> 5] = mhi;
}

__global__ void sp_down_table_tiles_kernel(const int* __restrict__ coords_out, int n_out, int pitch, SpConvGeom q,
                                           SpDims gi, const int* __restrict__ grid_in, int* __restrict__ nbr,
                                           unsigned* __restrict__ tmask)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kzy = (int)(e / pitch), o = (int)(e % pitch);
    const int ky = kzy % q.kh, kz = kzy / q.kh;
    int z = 0, y = 0, x0 = 0, b = 0;
    bool row_ok = false;
    if (o < n_out) {
        const int4 c = *reinterpret_cast<const int4*>(coords_out + 4 * (int64_t)o);
        b = c.x; z = c.y * q.sd - q.pd + kz; y = c.z * q.sh - q.ph + ky; x0 = c.w * q.sw - q.pw;
        row_ok = z >= 0 && z < gi.D && y >= 0 && y < gi.H;
    }
    const int64_t base = row_ok ? sp_cell(gi, b, z, y, 0) : 0;
    const int lane = threadIdx.x & 63;
    int v3[3];
    if (q.kw == 3) sp_probe3(grid_in, base, x0, gi.W, row_ok, v3);
    for (int kx = 0; kx < q.kw; ++kx) {
        const int x = x0 + kx;
        int v = -1;
        if (q.kw == 3) v = v3[kx];
        else if (row_ok && x >= 0 && x < gi.W) v = grid_in[base + x];
        const int k = kzy * q.kw + kx;
        nbr[(int64_t)k * pitch + o] = v;
        const unsigned long long bal = __ballot(v >= 0);
        if (lane == 0 && (bal & 0xffffffffull)) atomicOr(&tmask[o >> 5], 1u << k);
        if (lane == 32 && (bal >> 32)) atomicOr(&tmask[o >> 5], 1u << k);
    }
}

// ------------------------------------------------------------------ rows of a level grouped by their tap masks (round 5)
// The 64- and 128-channel submanifold layers run at 63-91 % of what the matrix cores sustain on the products they EXECUTE:
// every live (32-row tile, tap) pair in full, although only 57-59 % of its rows have a neighbour -- in raster order a tile
// executes 77 % of its 27 taps.  Which taps a tile executes depends only on WHICH rows share it, and the order of a level's rows
// is free inside the encoder.  So the rows of every WINDOW of W consecutive raster rows (W = 256 / 1024: a few x-lines of one
// plane, i.e. still spatial neighbours for the L2) are re-ordered by their own 27-bit neighbour mask: rows that lack the same
// taps end up in the same tiles, and whole-tile tap skips fire more often (level 3, measured on the bench's frames: 0.765 ->
// 0.687 executed at W = 256, 0.669 at W = 1024; the unreachable optimum, a sort over the whole batch, is 0.601).
// Two passes: the masks, one thread per row (the nine 3-cell probes of sp_table_rows27_kernel); then one workgroup per window: a
// bitonic sort of (mask, position) keys in LDS (stable, deterministic), coords_out[base + j] = coords_in[base + key_j.position]
// and the level's index grid renumbered in place.  (A first form did both in one 256-thread workgroup per window: 0.74 ms per
// level at W = 8192 -- the probes of 32 rows per thread in sequence; split: the probes run at full occupancy.)
// pass 1: one thread per row, its 27-bit neighbour mask (the probes of the table kernels, without the tables)
__global__ __launch_bounds__(256) void sp_row_masks_kernel(const int* __restrict__ coords, int n, SpDims g,
                                                           const int* __restrict__ grid, unsigned* __restrict__ masks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)i);
    const int xc = c.w - 1 < 0 ? 0 : (c.w - 1 > g.W - 3 ? g.W - 3 : c.w - 1);      // three cells at a clamped position
    const int shift = c.w - 1 - xc;                                                // (W >= 3 is checked by the entry)
    int v[9][3];
    bool ok[9];
#pragma unroll
    for (int kzy = 0; kzy < 9; ++kzy) {
        const int z = c.y + kzy / 3 - 1, y = c.z + kzy % 3 - 1;
        ok[kzy] = z >= 0 && z < g.D && y >= 0 && y < g.H;
        const int* p = grid + (ok[kzy] ? sp_cell(g, c.x, z, y, xc) : 0);           // unconditional loads: nine independent probes
        v[kzy][0] = p[0]; v[kzy][1] = p[1]; v[kzy][2] = p[2];
    }
    unsigned mask = 0u;
#pragma unroll
    for (int kzy = 0; kzy < 9; ++kzy)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x = c.w + kx - 1, idx = kx + shift;
            const int val = idx <= 0 ? v[kzy][0] : (idx == 1 ? v[kzy][1] : v[kzy][2]);
            if (ok[kzy] && x >= 0 && x < g.W && val >= 0) mask |= 1u << (kzy * 3 + kx);
        }
    masks[i] = mask;
}

// pass 2: one 1024-thread workgroup per window: bitonic sort of (mask, position) keys in LDS, then the permuted coords and the
// renumbered grid cells
template <int W>
__global__ __launch_bounds__(1024) void sp_mask_window_sort_kernel(const int* __restrict__ coords, int n, SpDims g,
                                                                   const unsigned* __restrict__ masks,
                                                                   int* __restrict__ grid, int* __restrict__ coords_out)
{
    constexpr int R = W / 1024;
    __shared__ unsigned long long key[W];
    const int base = blockIdx.x * W;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int loc = threadIdx.x + 1024 * r, i = base + loc;
        key[loc] = i < n ? (((unsigned long long)masks[i] << 16) | (unsigned)loc) : ~0ull;      // rows past n: stay at the end
    }
    __syncthreads();
    // W / 2 compare-exchanges per step, W / 2048 per thread, every lane busy: pair p -> (t, t + j) with t = 2 j (p / j) + p % j
    for (int k = 2; k <= W; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < (R > 1 ? R / 2 : 1); ++r) {
                const int p = threadIdx.x + 1024 * r;
                if (R == 1 && p >= W / 2) break;
                const int t = ((p & ~(j - 1)) << 1) | (p & (j - 1)), o = t + j;
                const unsigned long long a = key[t], b2 = key[o];
                const bool up = (t & k) == 0;
                if ((a > b2) == up) { key[t] = b2; key[o] = a; }
            }
            __syncthreads();
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int loc = threadIdx.x + 1024 * r, i = base + loc;
        if (i >= n) continue;
        const int src = base + (int)(key[loc] & 0xffffu);
        const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (int64_t)src);
        *reinterpret_cast<int4*>(coords_out + 4 * (int64_t)i) = c;
        grid[sp_cell(g, c.x, c.y, c.z, c.w)] = i;
    }
}

extern "C" int64_t al3d_sp_mask_window_sort_workspace_bytes(int n)
{
    return al3d_align((int64_t)(n > 0 ? n : 1) * 4, 256);
}

extern "C" int al3d_sp_mask_window_sort(const int* coords, int n, int B, int D, int H, int W, int* grid, int window,
                                        int* coords_out, void* workspace, void* stream)
{
    AL3D_REQUIRE(n >= 0 && B >= 1 && D >= 1 && H >= 1 && W >= 3, "al3d_sp_mask_window_sort: bad shape (W >= 3)");
    AL3D_REQUIRE(window == 1024 || window == 4096 || window == 8192 || window == 16384,
                 "al3d_sp_mask_window_sort: window must be 1024, 4096, 8192 or 16384 (got %d)", window);
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(coords && grid && coords_out && workspace && coords != coords_out,
                 "al3d_sp_mask_window_sort: null pointer / in-place call");
    AL3D_REQUIRE(((uintptr_t)coords & 15) == 0 && ((uintptr_t)coords_out & 15) == 0,
                 "al3d_sp_mask_window_sort: coords must be 16-byte aligned");
    SpDims g = {B, D, H, W};
    hipStream_t s = (hipStream_t)stream;
    unsigned* masks = (unsigned*)workspace;
    hipLaunchKernelGGL(sp_row_masks_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0, s, coords, n, g, grid, masks);
    const unsigned blocks = (unsigned)al3d_cdiv(n, window);
#define MS_LAUNCH(WV) hipLaunchKernelGGL(sp_mask_window_sort_kernel<WV>, dim3(blocks), dim3(1024), 0, s, coords, n, g, masks, grid, coords_out)
    if (window == 1024) MS_LAUNCH(1024);
    else if (window == 4096) MS_LAUNCH(4096);
    else if (window == 8192) MS_LAUNCH(8192);
    else MS_LAUNCH(16384);
#undef MS_LAUNCH
    AL3D_CHECK_LAUNCH("sp_mask_window_sort_kernel");
    return AL3D_OK;
}

// ------------------------------------------------------------------ the conv itself
// 32 output rows x COUT per 256-thread workgroup.  Thread -> one output channel and
// 32*COUT/256 rows; per kernel offset the 32 gathered input rows sit in LDS (broadcast
// float4 reads), the weight slice streams from L2 (coalesced over the output channel).
#define SP_TM 32

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sp_conv_kernel(const float* __restrict__ fin,
                                                      const int* __restrict__ nbr, int K,
                                                      const float* __restrict__ wgt,   // [K][CIN][COUT]
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift,
                                                      const float* __restrict__ residual, int relu,
                                                      float* __restrict__ fout, int n_out)
{
    constexpr int CPAD = (CIN + 3) & ~3;
    constexpr int GROUPS = 256 / COUT;          // row groups
    constexpr int RPT = SP_TM / GROUPS;         // rows per thread
    __shared__ __attribute__((aligned(16))) float a_s[SP_TM][CPAD];
    __shared__ int idx_s[SP_TM];
    const int tid = threadIdx.x;
    const int co = tid % COUT, rg = tid / COUT;
    const int row0 = blockIdx.x * SP_TM;
    float acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = 0.f;

    for (int k = 0; k < K; ++k) {
        int have = 0;
        if (tid < SP_TM) {
            const int row = row0 + tid;
            const int v = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
            idx_s[tid] = v;
            have = v >= 0;
        }
        if (!__syncthreads_or(have)) continue;       // nobody in this tile has offset k
        for (int e = tid; e < SP_TM * CPAD; e += 256) {
            const int r = e / CPAD, ci = e % CPAD;
            const int src = idx_s[r];
            a_s[r][ci] = (src >= 0 && ci < CIN) ? fin[(int64_t)src * CIN + ci] : 0.f;
        }
        __syncthreads();
        const float* wk = wgt + (int64_t)k * CIN * COUT + co;
#pragma unroll 4
        for (int ci = 0; ci < CPAD; ci += 4) {
            const float w0 = ci + 0 < CIN ? wk[(ci + 0) * COUT] : 0.f;
            const float w1 = ci + 1 < CIN ? wk[(ci + 1) * COUT] : 0.f;
            const float w2 = ci + 2 < CIN ? wk[(ci + 2) * COUT] : 0.f;
            const float w3 = ci + 3 < CIN ? wk[(ci + 3) * COUT] : 0.f;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(&a_s[rg * RPT + r][ci]);
                acc[r] = fmaf(a.x, w0, acc[r]);
                acc[r] = fmaf(a.y, w1, acc[r]);
                acc[r] = fmaf(a.z, w2, acc[r]);
                acc[r] = fmaf(a.w, w3, acc[r]);
            }
        }
        __syncthreads();
    }
    const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = row0 + rg * RPT + r;
        if (row >= n_out) continue;
        float v = acc[r] * sc + sh;
        if (residual) v += residual[(int64_t)row * COUT + co];
        if (relu) v = v <= 0.f ? 0.f : v;                   // NaN propagates, like torch.relu
        fout[(int64_t)row * COUT + co] = v;
    }
}

// sparse -> dense NHWC: out[b][y][x][c*D + z] = feat[row][c]  (SparseConvTensor.dense() then
// view(N, C*D, H, W), scn.py:387-390); `out` must be zero-filled by the caller.
__global__ void sp_to_dense_nhwc_kernel(const float* __restrict__ feat, const int* __restrict__ coords,
                                        int n, int C, SpDims g, float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * C) return;
    const int row = (int)(e / C), c = (int)(e % C);
    const int* q = coords + 4 * row;
    out[((((int64_t)q[0] * g.H + q[2]) * g.W + q[3]) * C + c) * g.D + q[1]] = feat[e];
}

// ------------------------------------------------------------------ C ABI
static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)al3d_cdiv(n > 0 ? n : 1, per); }

extern "C" int al3d_sp_fill_i32(int* buf, int64_t count, int value, void* stream)
{
    AL3D_REQUIRE(buf && count >= 0, "al3d_sp_fill_i32: bad arguments");
    if (value == 0 || value == -1) {
        if (hipMemsetAsync(buf, value == 0 ? 0 : 0xff, (size_t)count * 4, (hipStream_t)stream) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_sp_fill_i32: memset failed");
        return AL3D_OK;
    }
    return al3d_fail(AL3D_EINVAL, "al3d_sp_fill_i32: only 0 and -1 are supported");
}

extern "C" int al3d_sp_scatter_index(const int* coords, int n, int B, int D, int H, int W, int* grid,
                                     int mode, void* stream)
{
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(coords && grid && n > 0, "al3d_sp_scatter_index: bad arguments");
    AL3D_REQUIRE(((uintptr_t)coords & 15) == 0, "al3d_sp_scatter_index: coords must be 16-byte aligned");
    SpDims g = {B, D, H, W};
    hipLaunchKernelGGL(sp_scatter_index_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       coords, n, g, grid, mode);
    AL3D_CHECK_LAUNCH("sp_scatter_index_kernel");
    return AL3D_OK;
}

extern "C" int al3d_sp_subm_table(const int* coords, int n, int B, int D, int H, int W, const int* grid,
                                  int kd, int kh, int kw, int* nbr, void* stream)
{
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(coords && grid && nbr && n > 0, "al3d_sp_subm_table: bad arguments");
    AL3D_REQUIRE(kd % 2 == 1 && kh % 2 == 1 && kw % 2 == 1, "al3d_sp_subm_table: odd kernel sizes only");
    AL3D_REQUIRE(((uintptr_t)coords & 15) == 0, "al3d_sp_subm_table: coords must be 16-byte aligned");
    if (n == 0) return AL3D_OK;
    SpDims g = {B, D, H, W};
    hipLaunchKernelGGL(sp_subm_table_kernel, dim3(blocks_for((int64_t)n * kd * kh, 256)), dim3(256), 0,
                       (hipStream_t)stream, coords, n, g, grid, kd, kh, kw, nbr);
    AL3D_CHECK_LAUNCH("sp_subm_table_kernel");
    return AL3D_OK;
}

// multiple of 256: the table kernels run exact grids of 256-thread blocks whose waves cover 64 consecutive rows
extern "C" int al3d_sp_table_pitch(int n) { return (int)al3d_align(n > 0 ? n : 1, 256); }

extern "C" int al3d_sp_subm_table_tiles(const int* coords, int n, int B, int D, int H, int W, const int* grid,
                                        int kd, int kh, int kw, int* nbr, int pitch, unsigned* tile_mask,
                                        void* stream)
{
    AL3D_REQUIRE(n >= 0 && pitch == al3d_sp_table_pitch(n), "al3d_sp_subm_table_tiles: pitch must be al3d_sp_table_pitch(n)");
    AL3D_REQUIRE(nbr && tile_mask && (n == 0 || (coords && grid)), "al3d_sp_subm_table_tiles: null pointer");
    AL3D_REQUIRE(kd % 2 == 1 && kh % 2 == 1 && kw % 2 == 1 && kd * kh * kw <= 27,
                 "al3d_sp_subm_table_tiles: odd kernel sizes, at most 27 taps");
    AL3D_REQUIRE(((uintptr_t)coords & 15) == 0, "al3d_sp_subm_table_tiles: coords must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    SpDims g = {B, D, H, W};
    if (kd == 3 && kh == 3 && kw == 3 && W >= 3 && n > 0) {
        const SpConvGeom q1 = {3, 3, 3, 1, 1, 1, 1, 1, 1};
        hipLaunchKernelGGL(sp_table_rows27_kernel, dim3((unsigned)(pitch / 256)), dim3(256), 0, s, coords, n, pitch, q1, g, grid,
                           nbr, tile_mask);
        AL3D_CHECK_LAUNCH("sp_table_rows27_kernel");
        return AL3D_OK;
    }
    if (hipMemsetAsync(tile_mask, 0, (size_t)(pitch / 32) * 4, s) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "al3d_sp_subm_table_tiles: memset failed");
    hipLaunchKernelGGL(sp_subm_table_tiles_kernel, dim3((unsigned)((int64_t)pitch * kd * kh / 256)), dim3(256), 0, s,
                       coords, n, pitch, g, grid, kd, kh, kw, nbr, tile_mask);
    AL3D_CHECK_LAUNCH("sp_subm_table_tiles_kernel");
    return AL3D_OK;
}

extern "C" int al3d_sp_down_claim(const int* coords_in, int n_in, const int* ksize, const int* stride,
                                  const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                                  int* coords_out, int* counter, int cap, void* stream)
{
    if (n_in == 0) return AL3D_OK;
    AL3D_REQUIRE(coords_in && ksize && stride && pad && grid_out && coords_out && counter,
                 "al3d_sp_down_claim: null pointer");
    SpConvGeom q = {ksize[0], ksize[1], ksize[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2]};
    SpDims go = {B, OD, OH, OW};
    hipLaunchKernelGGL(sp_down_claim_kernel, dim3(blocks_for(n_in, 256)), dim3(256), 0,
                       (hipStream_t)stream, coords_in, n_in, q, go, grid_out, coords_out, counter, cap);
    AL3D_CHECK_LAUNCH("sp_down_claim_kernel");
    return AL3D_OK;
}

static int64_t sp_down_cells(int B, int OD, int OH, int OW, bool blocked)
{
    return blocked ? (int64_t)B * ((OH + 7) / 8) * ((OW + 7) / 8) * OD * 64 : (int64_t)B * OD * OH * OW;
}

static int64_t sp_down_sites_ws(int B, int OD, int OH, int OW, bool blocked)
{
    const int64_t words = (sp_down_cells(B, OD, OH, OW, blocked) + 31) / 32;
    return al3d_align(words * 32, 256) + 3 * al3d_align(words * 4, 256) + al3d_scan_workspace_bytes(words);
}

extern "C" int64_t al3d_sp_down_sites_workspace_bytes(int B, int OD, int OH, int OW)
{
    return sp_down_sites_ws(B, OD, OH, OW, false);
}

extern "C" int64_t al3d_sp_down_sites_blocked_workspace_bytes(int B, int OD, int OH, int OW)
{
    return sp_down_sites_ws(B, OD, OH, OW, true);
}

template <bool BLOCKED>
static int sp_down_sites_impl(const int* coords_in, int n_in, const int* ksize, const int* stride,
                              const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                              int* coords_out, int* counter, int cap, void* workspace, void* stream)
{
    AL3D_REQUIRE(ksize && stride && pad && grid_out && coords_out && counter && workspace,
                 "al3d_sp_down_sites: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t cells = sp_down_cells(B, OD, OH, OW, BLOCKED);
    AL3D_REQUIRE(cells > 0 && cells < (1LL << 31), "al3d_sp_down_sites: bad grid");
    AL3D_REQUIRE(((uintptr_t)coords_out & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                 "al3d_sp_down_sites: coords_out / workspace must be 16-byte aligned");
    const int64_t words = (cells + 31) / 32;
    const int64_t fb = al3d_align(words * 32, 256), wb = al3d_align(words * 4, 256);
    unsigned char* flags = (unsigned char*)workspace;
    unsigned* bits = (unsigned*)(flags + fb);
    int* cnt = (int*)(flags + fb + wb);
    int* wscan = (int*)(flags + fb + 2 * wb);
    void* scan_ws = flags + fb + 3 * wb;
    if (n_in > 0) {
        AL3D_REQUIRE(coords_in, "al3d_sp_down_sites: null coords");
        if (hipMemsetAsync(flags, 0, (size_t)words * 32, s) != hipSuccess)
            return al3d_fail(AL3D_ELAUNCH, "al3d_sp_down_sites: memset failed");
        SpConvGeom q = {ksize[0], ksize[1], ksize[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2]};
        SpDims go = {B, OD, OH, OW};
        hipLaunchKernelGGL(sp_down_mark_kernel<BLOCKED>, dim3(blocks_for(n_in, 256)), dim3(256), 0, s, coords_in, n_in,
                           q, go, flags);
        hipLaunchKernelGGL(sp_down_pack_kernel, dim3(blocks_for(words, 256)), dim3(256), 0, s, flags, words, bits, cnt);
        int rc = al3d_exclusive_scan_i32(cnt, wscan, words, scan_ws, s);
        if (rc) return rc;
        hipLaunchKernelGGL(sp_down_assign_kernel<BLOCKED>, dim3(blocks_for(words, 256)), dim3(256), 0, s, bits, wscan,
                           words, go, grid_out, coords_out, cap);
        hipLaunchKernelGGL(sp_count_tail_kernel, dim3(1), dim3(64), 0, s, bits, wscan, words, counter);
    } else if (hipMemsetAsync(counter, 0, 4, s) != hipSuccess) {
        return al3d_fail(AL3D_ELAUNCH, "al3d_sp_down_sites: memset failed");
    }
    AL3D_CHECK_LAUNCH("sp_down_sites");
    return AL3D_OK;
}

extern "C" int al3d_sp_down_sites(const int* coords_in, int n_in, const int* ksize, const int* stride,
                                  const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                                  int* coords_out, int* counter, int cap, void* workspace, void* stream)
{
    return sp_down_sites_impl<false>(coords_in, n_in, ksize, stride, pad, B, OD, OH, OW, grid_out, coords_out, counter, cap,
                                     workspace, stream);
}

extern "C" int al3d_sp_down_sites_blocked(const int* coords_in, int n_in, const int* ksize, const int* stride,
                                          const int* pad, int B, int OD, int OH, int OW, int* grid_out,
                                          int* coords_out, int* counter, int cap, void* workspace, void* stream)
{
    return sp_down_sites_impl<true>(coords_in, n_in, ksize, stride, pad, B, OD, OH, OW, grid_out, coords_out, counter, cap,
                                    workspace, stream);
}

extern "C" int al3d_sp_down_table(const int* coords_out, int n_out, const int* ksize, const int* stride,
                                  const int* pad, int B, int ID, int IH, int IW, const int* grid_in,
                                  int* nbr, void* stream)
{
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(coords_out && ksize && stride && pad && grid_in && nbr, "al3d_sp_down_table: null pointer");
    AL3D_REQUIRE(((uintptr_t)coords_out & 15) == 0, "al3d_sp_down_table: coords_out must be 16-byte aligned");
    SpConvGeom q = {ksize[0], ksize[1], ksize[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2]};
    SpDims gi = {B, ID, IH, IW};
    hipLaunchKernelGGL(sp_down_table_kernel, dim3(blocks_for((int64_t)n_out * q.kd * q.kh, 256)),
                       dim3(256), 0, (hipStream_t)stream, coords_out, n_out, q, gi, grid_in, nbr);
    AL3D_CHECK_LAUNCH("sp_down_table_kernel");
    return AL3D_OK;
}

extern "C" int al3d_sp_down_table_tiles(const int* coords_out, int n_out, const int* ksize, const int* stride,
                                        const int* pad, int B, int ID, int IH, int IW, const int* grid_in,
                                        int* nbr, int pitch, unsigned* tile_mask, void* stream)
{
    AL3D_REQUIRE(n_out >= 0 && pitch == al3d_sp_table_pitch(n_out),
                 "al3d_sp_down_table_tiles: pitch must be al3d_sp_table_pitch(n_out)");
    AL3D_REQUIRE(ksize && stride && pad && nbr && tile_mask && (n_out == 0 || (coords_out && grid_in)),
                 "al3d_sp_down_table_tiles: null pointer");
    AL3D_REQUIRE(((uintptr_t)coords_out & 15) == 0, "al3d_sp_down_table_tiles: coords_out must be 16-byte aligned");
    AL3D_REQUIRE(ksize[0] * ksize[1] * ksize[2] <= 27, "al3d_sp_down_table_tiles: at most 27 taps");
    hipStream_t s = (hipStream_t)stream;
    SpConvGeom q = {ksize[0], ksize[1], ksize[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2]};
    SpDims gi = {B, ID, IH, IW};
    if (q.kd == 3 && q.kh == 3 && q.kw == 3 && IW >= 3 && n_out > 0) {
        hipLaunchKernelGGL(sp_table_rows27_kernel, dim3((unsigned)(pitch / 256)), dim3(256), 0, s, coords_out, n_out, pitch, q, gi,
                           grid_in, nbr, tile_mask);
        AL3D_CHECK_LAUNCH("sp_table_rows27_kernel");
        return AL3D_OK;
    }
    if (hipMemsetAsync(tile_mask, 0, (size_t)(pitch / 32) * 4, s) != hipSuccess)
        return al3d_fail(AL3D_ELAUNCH, "al3d_sp_down_table_tiles: memset failed");
    hipLaunchKernelGGL(sp_down_table_tiles_kernel, dim3((unsigned)((int64_t)pitch * q.kd * q.kh / 256)), dim3(256), 0, s,
                       coords_out, n_out, pitch, q, gi, grid_in, nbr, tile_mask);
    AL3D_CHECK_LAUNCH("sp_down_table_tiles_kernel");
    return AL3D_OK;
}

#define SP_DISPATCH(CI, CO)                                                                        \
    if (cin == CI && cout == CO) {                                                                 \
        hipLaunchKernelGGL((sp_conv_kernel<CI, CO>), dim3(blocks_for(n_out, SP_TM)), dim3(256), 0, s, \
                           fin, nbr, K, wgt, scale, shift, residual, relu, fout, n_out);           \
        AL3D_CHECK_LAUNCH("sp_conv_kernel");                                                       \
        return AL3D_OK;                                                                            \
    }

extern "C" int al3d_sp_conv_f32(const float* fin, const int* nbr, int K, const float* wgt, int cin,
                                int cout, const float* scale, const float* shift,
                                const float* residual, int relu, float* fout, int n_out, void* stream)
{
    AL3D_REQUIRE(K >= 1 && n_out >= 0, "al3d_sp_conv_f32: bad sizes");
    if (n_out == 0) return AL3D_OK;
    AL3D_REQUIRE(fin && nbr && wgt && fout, "al3d_sp_conv_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    SP_DISPATCH(5, 16) SP_DISPATCH(4, 16) SP_DISPATCH(16, 16) SP_DISPATCH(16, 32) SP_DISPATCH(32, 32)
    SP_DISPATCH(32, 64) SP_DISPATCH(64, 64) SP_DISPATCH(64, 128) SP_DISPATCH(128, 128)
    return al3d_fail(AL3D_EINVAL, "al3d_sp_conv_f32: unsupported channel pair %d -> %d", cin, cout);
}

extern "C" int al3d_sp_to_dense_nhwc(const float* feat, const int* coords, int n, int C, int B, int D,
                                     int H, int W, float* out, void* stream)
{
    if (n == 0) return AL3D_OK;
    AL3D_REQUIRE(feat && coords && out && n > 0, "al3d_sp_to_dense_nhwc: bad arguments");
    SpDims g = {B, D, H, W};
    hipLaunchKernelGGL(sp_to_dense_nhwc_kernel, dim3(blocks_for((int64_t)n * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, coords, n, C, g, out);
    AL3D_CHECK_LAUNCH("sp_to_dense_nhwc_kernel");
    return AL3D_OK;
}
