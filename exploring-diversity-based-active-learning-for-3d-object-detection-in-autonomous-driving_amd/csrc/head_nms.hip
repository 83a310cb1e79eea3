// Anchor-head post-processing on device: box decode, sigmoid score + class max,
// score threshold, top-k, rotated (exact polygon) NMS and the per-frame assembly.
//
// Reference: MultiGroupHead.predict / get_task_detections
// (det3d/models/bbox_heads/mg_head.py:697-803,805-1085), second_box_decode
// (det3d/core/bbox/box_torch_ops.py:80-148), rotate_nms (box_torch_ops.py:528-550),
// rotate_nms_cc (det3d/ops/nms/nms_cpu.py:34-45) and rotate_non_max_suppression_cpu
// (det3d/ops/nms/nms_cpu.h:73-168).  The reference round-trips every (task, sample) pair
// through the host (D2H -> boost::geometry loop -> H2D); here one workgroup per
// (sample, task) does select + sort + NMS entirely in LDS.
//
// Selection order is made deterministic: higher score first, equal scores -> lower anchor
// index (torch.topk / numpy argsort leave ties unspecified).
#include "al3d_common.h"

#define HN_THREADS 1024
#define HN_MAXK 1024            // nms_pre_max_size <= 1024
#ifndef HN_G
#define HN_G 8                  // tentative survivors per super-round of the greedy loop (<= 16)
#endif
#define HN_GB (HN_G <= 8 ? 8 : 16)                  // verdict bits kept per candidate
#define HN_GPW (32 / HN_GB)                          // candidates per verdict word
#define HN_GSH(j) (((j) % HN_GPW) * HN_GB)
#define HN_GMASK ((1u << HN_GB) - 1u)

struct HeadTask {
    const float* anchors;   // [A, 9] x y z w l h vx vy r
    int A;                  // anchors = H*W*na
    int na, nc;             // anchors per location, classes
    int box_off, cls_off;   // channel offsets inside the fused head output
    int label_off;          // class-id offset of this task in the merged label space
};

struct HeadParams {
    const float* hout;      // [B, H*W, CH] fused head output (NHWC)
    int B, HW, CH, ntasks;
    float score_thresh, iou_thresh;
    int pre_max, post_max;
    float range[6];         // post_center_limit_range
    HeadTask task[8];
    // outputs, [B, ntasks, post_max, *]
    float* boxes;           // 9 floats
    float* scores;
    int* labels;
    int* counts;            // [B, ntasks]
    unsigned* sbits;        // workspace: per (sample, task, anchor) score bit pattern, 0 = below threshold
    int64_t task_soff[8];   // offset of task t inside one sample's block of sbits
    int64_t sample_stride;  // sum of A over the tasks
    int order[8];           // tasks by anchor count, largest first: the long problems of a launch start first
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// second_box_decode with encode_angle_to_vector=True, smooth_dim=False, norm_velo=False
__device__ __forceinline__ void decode_box(const float* __restrict__ t, const float* __restrict__ a,
                                           float* __restrict__ o)
{
    const float xa = a[0], ya = a[1], za = a[2], wa = a[3], la = a[4], ha = a[5], vxa = a[6], vya = a[7],
                ra = a[8];
    const float diag = sqrtf(la * la + wa * wa);
    o[0] = t[0] * diag + xa;
    o[1] = t[1] * diag + ya;
    o[2] = t[2] * ha + za;
    o[3] = expf(t[3]) * wa;
    o[4] = expf(t[4]) * la;
    o[5] = expf(t[5]) * ha;
    o[6] = t[6] + vxa;
    o[7] = t[7] + vya;
    o[8] = atan2f(t[9] + sinf(ra), t[8] + cosf(ra));
}

// corners of (x, y, w, l, r): unit square (0,0),(0,1),(1,1),(1,0) minus 0.5, scaled by (w,l),
// rotated by [[cos,-sin],[sin,cos]] applied as row-vector @ R^T ... the reference's
// rotation_2d: einsum("aij,jka->aik", points, [[c,-s],[s,c]])  => x' = x*c + y*s, y' = -x*s + y*c
__device__ __forceinline__ void box_corners(float x, float y, float w, float l, float r, float* cx,
                                            float* cy)
{
    const float c = cosf(r), s = sinf(r);
    const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float px = ux[k] * w, py = uy[k] * l;
        cx[k] = px * c + py * s + x;
        cy[k] = -px * s + py * c + y;
    }
}

// area of the intersection of two convex quadrilaterals (Sutherland-Hodgman clip + shoelace)
__device__ float quad_intersection_area(const float* ax, const float* ay, const float* bx, const float* by)
{
    float px[10], py[10], qx[10], qy[10];
    int n = 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { px[k] = ax[k]; py[k] = ay[k]; }
    // orientation of the clip polygon
    float barea = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int k2 = (k + 1) & 3; barea += bx[k] * by[k2] - bx[k2] * by[k]; }
    const float sgn = barea >= 0.f ? 1.f : -1.f;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const int e2 = (e + 1) & 3;
        const float ex = bx[e2] - bx[e], ey = by[e2] - by[e];
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const int k2 = k + 1 == n ? 0 : k + 1;
            const float d1 = sgn * (ex * (py[k] - by[e]) - ey * (px[k] - bx[e]));
            const float d2 = sgn * (ex * (py[k2] - by[e]) - ey * (px[k2] - bx[e]));
            if (d1 >= 0.f) { qx[m] = px[k]; qy[m] = py[k]; ++m; }
            if ((d1 >= 0.f) != (d2 >= 0.f)) {
                const float tt = d1 / (d1 - d2);
                qx[m] = px[k] + tt * (px[k2] - px[k]);
                qy[m] = py[k] + tt * (py[k2] - py[k]);
                ++m;
            }
        }
        n = m;
        for (int k = 0; k < n; ++k) { px[k] = qx[k]; py[k] = qy[k]; }
    }
    if (n < 3) return 0.f;
    float area = 0.f;
    for (int k = 0; k < n; ++k) { const int k2 = k + 1 == n ? 0 : k + 1; area += px[k] * py[k2] - px[k2] * py[k]; }
    return 0.5f * fabsf(area);
}

// Upper bound of the intersection of rectangle P (corners p, area pa) with rectangle Q from Q's bounding box in P's OWN frame
// (axes = P's edges c0->c3 and c0->c1, unnormalised: coordinates scale with the squared edge lengths, so no root is taken):
// inter <= ow * oh / pa.  Returns false when that bound already keeps IoU under `lim` (or the boxes are apart along one of
// P's axes), i.e. when the exact clip cannot end in a suppression.  Exact-safe: used with a 0.1 % margin folded into `lim`.
__device__ __forceinline__ bool obb_bound(const float* px, const float* py, float pa, const float* qx, const float* qy, float qa,
                                          float lim)
{
    const float ux = px[3] - px[0], uy = py[3] - py[0], vx = px[1] - px[0], vy = py[1] - py[0];
    const float mx = 0.5f * (px[0] + px[2]), my = 0.5f * (py[0] + py[2]);
    const float hu = 0.5f * (ux * ux + uy * uy), hv = 0.5f * (vx * vx + vy * vy);
    float amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = qx[k] - mx, dy = qy[k] - my;
        const float a = dx * ux + dy * uy, b = dx * vx + dy * vy;
        amin = fminf(amin, a); amax = fmaxf(amax, a); bmin = fminf(bmin, b); bmax = fmaxf(bmax, b);
    }
    const float ow = fminf(amax, hu) - fmaxf(amin, -hu), oh = fminf(bmax, hv) - fmaxf(bmin, -hv);
    if (!(ow > 0.f && oh > 0.f)) return !(ow <= 0.f || oh <= 0.f);     // apart along an axis -> no clip; NaN -> keep the clip
    const float bound = 1.001f * (ow * oh) / pa;                       // >= the true intersection area
    return bound >= lim * (pa + qa - bound) || !(bound == bound);
}

__device__ __forceinline__ float quad_area(const float* x, const float* y)
{
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int k2 = (k + 1) & 3; a += x[k] * y[k2] - x[k2] * y[k]; }
    return 0.5f * fabsf(a);
}

// Pass 0, whole GPU: score of every anchor = max over classes of sigmoid(cls) -> its bit pattern
// (0 when below the threshold or NaN).  The select passes of head_nms_kernel then stream this
// compact array (coalesced, L2-resident) instead of re-reading the 944-byte-strided head output
// five times from one CU (measured: 2.0 ms -> see DESIGN.md).
__global__ __launch_bounds__(256) void head_score_kernel(HeadParams p)
{
    const int b = blockIdx.z, t = blockIdx.y;
    const HeadTask tk = p.task[t];
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= tk.A) return;
    const float* hb = p.hout + (int64_t)b * p.HW * p.CH;
    const int loc = a / tk.na, j = a % tk.na;
    const float* c = hb + (int64_t)loc * p.CH + tk.cls_off + j * tk.nc;
    float best = sigmoidf_(c[0]);
    for (int q = 1; q < tk.nc; ++q) { const float s = sigmoidf_(c[q]); if (s > best) best = s; }
    p.sbits[(int64_t)b * p.sample_stride + p.task_soff[t] + a] = best >= p.score_thresh ? __float_as_uint(best) : 0u;
}

// The same pre-pass with the class logits of 256 locations staged through LDS: the fused head output keeps all class
// logits of a location in ONE contiguous window [cls_lo, cls_lo + cls_w) of its CH-float record (bbox_heads.py orders it
// [all box regressions | all class logits]), so the workgroup fetches 256 x cls_w floats with consecutive lanes on
// consecutive addresses (2-3 cache lines per wave-load instead of 64 with one anchor per thread) and then one thread
// per location takes the sigmoid / max of every (task, anchor).  Same sigmoid, same order of comparisons: same bits.
#define HS_LOCS 64
#define HS_MAXW 128
__global__ __launch_bounds__(256) void head_score_rows_kernel(HeadParams p, int cls_lo, int cls_w)
{
    extern __shared__ float lg[];                                      // HS_LOCS x (cls_w + 1) floats
    const int b = blockIdx.y, loc0 = blockIdx.x * HS_LOCS;
    const int nloc = p.HW - loc0 < HS_LOCS ? p.HW - loc0 : HS_LOCS;
    const float* hb = p.hout + ((int64_t)b * p.HW + loc0) * p.CH + cls_lo;
    const int pitch = cls_w + 1;
    if (((cls_lo | cls_w | p.CH) & 3) == 0) {                          // 16-byte pieces (the shipped head: 180 | 56 of 236)
        const int q4 = cls_w >> 2;
        for (int e = threadIdx.x; e < nloc * q4; e += 256) {
            const int l = e / q4, k = (e - l * q4) * 4;
            const float4 v = *reinterpret_cast<const float4*>(hb + (int64_t)l * p.CH + k);
            float* d = lg + l * pitch + k;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < nloc * cls_w; e += 256) {
            const int l = e / cls_w, k = e - l * cls_w;
            lg[l * pitch + k] = hb[(int64_t)l * p.CH + k];
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < nloc * p.ntasks; w += 256) {          // one (task, location) per thread, locations fastest
        const int t = w / nloc, l = w - t * nloc;
        const HeadTask tk = p.task[t];
        unsigned* o = p.sbits + (int64_t)b * p.sample_stride + p.task_soff[t] + (int64_t)(loc0 + l) * tk.na;
        for (int j = 0; j < tk.na; ++j) {
            const float* c = lg + l * pitch + (tk.cls_off - cls_lo) + j * tk.nc;
            float best = sigmoidf_(c[0]);
            for (int q = 1; q < tk.nc; ++q) { const float s2 = sigmoidf_(c[q]); if (s2 > best) best = s2; }
            o[j] = best >= p.score_thresh ? __float_as_uint(best) : 0u;
        }
    }
}

__global__ __launch_bounds__(HN_THREADS) void head_nms_kernel(HeadParams p)
{
    // 768 problems on 512 resident workgroups (two per CU): the ones that start late should be the short ones
    const int t = p.order[blockIdx.x / p.B], b = blockIdx.x % p.B;
    const HeadTask tk = p.task[t];
    const int tid = threadIdx.x;
    const float* hb = p.hout + (int64_t)b * p.HW * p.CH;
    const unsigned* __restrict__ sb = p.sbits + (int64_t)b * p.sample_stride + p.task_soff[t];

    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need, s_sel_cnt, s_eq_seen;
    __shared__ unsigned long long key_s[HN_MAXK];       // (score bits << 32) | ~anchor
    __shared__ float cx_s[HN_MAXK][4], cy_s[HN_MAXK][4];
    // per-wave histograms of the select passes: 16 x 256 words in cx_s's storage (free until the decode phase)
    unsigned (*hist_w)[256] = reinterpret_cast<unsigned (*)[256]>(&cx_s[0][0]);
    static_assert(sizeof(cx_s) >= (HN_THREADS / 64) * 256 * sizeof(unsigned), "histograms must fit cx_s");
    __shared__ float sb_s[HN_MAXK][4];                  // standup box x1,y1,x2,y2
    __shared__ float area_s[HN_MAXK];
    __shared__ int keep_s[128];
    __shared__ int s_kept;

    // score of anchor a: max over classes of sigmoid(cls); label = first argmax
    auto anchor_score = [&](int a, int& label) -> float {
        const int loc = a / tk.na, j = a % tk.na;
        const float* c = hb + (int64_t)loc * p.CH + tk.cls_off + j * tk.nc;
        float best = sigmoidf_(c[0]);
        label = 0;
        for (int q = 1; q < tk.nc; ++q) { const float s = sigmoidf_(c[q]); if (s > best) { best = s; label = q; } }
        return best;
    };

    // ---- radix select of the pre_max largest scores among those >= score_thresh
    // scores are in (0,1): positive floats order like their bit patterns.
    const int K = p.pre_max < HN_MAXK ? p.pre_max : HN_MAXK;
    unsigned prefix = 0, need = K;
    if (tid == 0) { s_sel_cnt = 0; s_eq_seen = 0; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int q = tid; q < 256; q += HN_THREADS) hist[q] = 0;
        if (pass > 0)
            for (int q = tid; q < (HN_THREADS / 64) * 256; q += HN_THREADS) (&hist_w[0][0])[q] = 0;
        __syncthreads();
        // four chunks of anchors per trip: the four loads are issued together (one L2 round trip per trip instead of
        // one per chunk -- the ballots below keep the compiler from overlapping trips by itself)
        for (int a00 = 0; a00 < tk.A; a00 += 4 * HN_THREADS) {
          unsigned bits4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
              const int a = a00 + u * HN_THREADS + tid;
              bits4[u] = a < tk.A ? sb[a] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned bits = bits4[u];
            int bin = -1;
            if (bits && (pass == 0 || (bits >> (shift + 8)) == (prefix >> (shift + 8)))) bin = (int)((bits >> shift) & 255u);
            // scores of one task share their leading bits (all in (0.1, 1)): a plain per-thread atomicAdd serialises
            // ~A deep on one LDS word.  Aggregate per wave: one atomic per distinct bin and wave.
            unsigned long long pending = __ballot(bin >= 0);
            if (pass == 0) {
                while (pending) {
                    const int leader = (int)__builtin_ctzll(pending);
                    const int lb = __builtin_amdgcn_readlane(bin, leader);
                    const unsigned long long same = __ballot(bin == lb);
                    if ((tid & 63) == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
                    pending &= ~same;
                }
            } else if (pending) {
                // Later bytes are mantissa bits: a wave's 64 entries spread over ~50 distinct bins, and the aggregation loop
                // above would run once per distinct bin (it was most of this kernel's time on a pool where every anchor
                // passes the score threshold).  One aggregated round for the first lane's bin (covers the massive-tie
                // case), the other lanes add to their wave's private histogram with plain LDS atomics.
                const int leader = (int)__builtin_ctzll(pending);
                const int lb = __builtin_amdgcn_readlane(bin, leader);
                const unsigned long long same = __ballot(bin == lb);
                if ((tid & 63) == leader) atomicAdd(&hist_w[tid >> 6][lb], (unsigned)__popcll(same));
                if (bin >= 0 && bin != lb) atomicAdd(&hist_w[tid >> 6][bin], 1u);
            }
          }
        }
        __syncthreads();
        if (pass > 0) {                                  // fold the per-wave histograms
            for (int q = tid; q < 256; q += HN_THREADS) {
                unsigned tot = 0;
#pragma unroll
                for (int w = 0; w < HN_THREADS / 64; ++w) tot += hist_w[w][q];
                hist[q] = tot;
            }
            __syncthreads();
        }
        if (tid == 0) {
            unsigned acc = 0; int bin = 255;
            for (; bin >= 0; --bin) { if (acc + hist[bin] >= need) break; acc += hist[bin]; }
            if (bin < 0) { bin = 0; s_need = 0xffffffffu; }   // fewer than `need` candidates: take all
            else s_need = need - acc;
            s_prefix = prefix | ((unsigned)bin << shift);
        }
        __syncthreads();
        prefix = s_prefix;
        if (s_need == 0xffffffffu) { prefix = 0; need = 0xffffffffu; break; }
        need = s_need;
        __syncthreads();
    }
    // prefix = bit pattern of the K-th largest score (or 0 = take everything); `need` = how many
    // entries equal to it still fit (ties -> lower anchor index first).
    __syncthreads();
    const unsigned thr_bits = prefix;
    // Entries strictly above the threshold are taken in any order (the sort below orders them by (score, anchor));
    // entries EQUAL to it are admitted in ascending anchor order until `need` of them fit: they are collected into
    // a small list (exact score ties are rare), ranked by anchor, and the first `need` are taken.  No barrier per
    // chunk of anchors.  More than HN_MAXK ties: the chunked path below (three barriers per 1024 anchors).
    __shared__ unsigned s_eq_cnt;
    __shared__ int eq_list[HN_MAXK];
    if (tid == 0) s_eq_cnt = 0;
    __syncthreads();
    for (int a00 = tid; a00 < tk.A; a00 += 4 * HN_THREADS) {
      unsigned bits4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bits4[u] = a00 + u * HN_THREADS < tk.A ? sb[a00 + u * HN_THREADS] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = a00 + u * HN_THREADS;
        const unsigned bits = bits4[u];
        if (!bits) continue;
        if (need == 0xffffffffu || bits > thr_bits) {
            const unsigned slot = atomicAdd(&s_sel_cnt, 1u);
            if (slot < HN_MAXK) key_s[slot] = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)a);
        } else if (bits == thr_bits) {
            const unsigned e = atomicAdd(&s_eq_cnt, 1u);
            if (e < HN_MAXK) eq_list[e] = a;
        }
      }
    }
    __syncthreads();
    const unsigned neq = s_eq_cnt;
    if (neq <= HN_MAXK) {
        // rank of an equal entry = number of equal entries with a smaller anchor (anchors are distinct)
        if (tid < (int)neq && need != 0xffffffffu) {
            const int a = eq_list[tid];
            unsigned rank = 0;
            for (unsigned q = 0; q < neq; ++q) rank += eq_list[q] < a ? 1u : 0u;
            if (rank < need) {
                const unsigned slot = atomicAdd(&s_sel_cnt, 1u);
                if (slot < HN_MAXK) key_s[slot] = ((unsigned long long)thr_bits << 32) | (unsigned)(0xffffffffu - (unsigned)a);
            }
        }
        __syncthreads();
    } else {
    // more ties than the list holds: admit them chunk by chunk in anchor order
    for (int a0 = 0; a0 < tk.A; a0 += HN_THREADS) {
        const int a = a0 + tid;
        bool take = false, eq = false;
        unsigned bits = 0;
        if (a < tk.A) {
            bits = sb[a];
            if (bits && bits == thr_bits) eq = true;
        }
        __shared__ unsigned eq_wave[HN_THREADS / 64];
        const unsigned long long em = __ballot(eq);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) eq_wave[wv] = (unsigned)__popcll(em);
        __syncthreads();
        if (eq) {
            unsigned rank = s_eq_seen + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; ++w) rank += eq_wave[w];
            if (rank < need) take = true;
        }
        __syncthreads();
        if (tid == 0) { unsigned tot = 0; for (int w = 0; w < HN_THREADS / 64; ++w) tot += eq_wave[w]; s_eq_seen += tot; }
        if (take) {
            const unsigned slot = atomicAdd(&s_sel_cnt, 1u);
            if (slot < HN_MAXK) key_s[slot] = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)a);
        }
        __syncthreads();
    }
    }
#if defined(AL3D_NMS_STOP) && AL3D_NMS_STOP == 1
    return;
#endif
    const int n = (int)(s_sel_cnt < (unsigned)K ? s_sel_cnt : (unsigned)K);
    // ---- bitonic sort of the (<= 1024) keys, descending
    for (int q = n + tid; q < HN_MAXK; q += HN_THREADS) key_s[q] = 0ull;
    __syncthreads();
    for (int size = 2; size <= HN_MAXK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int i = tid, j = i ^ stride;
            if (j > i) {
                const bool desc = (i & size) == 0;
                const unsigned long long x = key_s[i], y = key_s[j];
                if (desc ? x < y : x > y) { key_s[i] = y; key_s[j] = x; }
            }
            __syncthreads();
        }
    }
    // ---- decode the selected boxes, corners, standup boxes
    float box[9];
    int my_label = 0;
    float my_score = 0.f;
    if (tid < n) {
        const unsigned long long key = key_s[tid];
        const int a = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
        my_score = __uint_as_float((unsigned)(key >> 32));
        int lab;
        anchor_score(a, lab);
        my_label = lab;
        const int loc = a / tk.na, j = a % tk.na;
        decode_box(hb + (int64_t)loc * p.CH + tk.box_off + j * 10, tk.anchors + (int64_t)a * 9, box);
        box_corners(box[0], box[1], box[3], box[4], box[8], cx_s[tid], cy_s[tid]);
        float x1 = cx_s[tid][0], x2 = x1, y1 = cy_s[tid][0], y2 = y1;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            x1 = fminf(x1, cx_s[tid][k]); x2 = fmaxf(x2, cx_s[tid][k]);
            y1 = fminf(y1, cy_s[tid][k]); y2 = fmaxf(y2, cy_s[tid][k]);
        }
        sb_s[tid][0] = x1; sb_s[tid][1] = y1; sb_s[tid][2] = x2; sb_s[tid][3] = y2;
        area_s[tid] = quad_area(cx_s[tid], cy_s[tid]);
    }
#if defined(AL3D_NMS_STOP) && AL3D_NMS_STOP == 2
    return;
#endif
    // ---- greedy rotated NMS; only the first post_max survivors are needed.
    // The sequential rule -- the best live candidate survives and suppresses what overlaps it, repeat -- is evaluated HN_G
    // candidates at a time: a super-round takes the next HN_G live candidates t_0 < t_1 < .. in rank order as TENTATIVE
    // survivors, tests every later live candidate against each of them (the tentatives against each other included),
    // and then resolves in rank order: t_g stands iff no STANDING t_h (h < g) suppresses it, and a candidate dies iff a
    // standing tentative suppresses it.  That is the sequential outcome exactly (between two consecutive tentatives there
    // is no other live candidate; verdicts of tentatives that fall are discarded), with one exact-clip phase -- the part
    // whose latency bounds this kernel, ~10 us per phase -- per HN_G candidates instead of per survivor.
    // The exact clips of a super-round are COMPACTED: the (candidate, tentative) pairs that survive the cheap tests (standup
    // boxes overlap, the three IoU bounds allow a suppression) are appended to a work list and the first c threads clip them
    // -- ceil(c / 64) waves run the ~3k-instruction clip instead of every wave that owns such a candidate.  Same clips,
    // same arithmetic, same decisions as one survivor at a time.
    const int post = p.post_max < 128 ? p.post_max : 128;
    int* work_s = eq_list;                             // selection is over: its tie list's storage holds the work list
    __shared__ unsigned long long live_m[HN_THREADS / 64];
    __shared__ unsigned supm[HN_MAXK / HN_GPW];        // per candidate: HN_GB bits, bit g = "tentative g suppresses it"
    __shared__ int s_wcnt;
    for (int q = tid; q < HN_MAXK / HN_GPW; q += HN_THREADS) supm[q] = 0u;
    if (tid == 0) s_wcnt = 0;
    bool dead = tid >= n;                              // this thread's candidate is suppressed / kept / absent
    int kept = 0;
    const float lim = 0.999f * p.iou_thresh;
    while (kept < post) {
        const unsigned long long live = __ballot(!dead);
        if ((tid & 63) == 0) live_m[tid >> 6] = live;
        __syncthreads();                               // live masks (and the previous super-round's clean-up) visible
        int tg[HN_G];
        int nt = 0;
#pragma unroll
        for (int g = 0; g < HN_G; ++g) tg[g] = HN_MAXK;
        for (int w = 0; w < HN_THREADS / 64 && nt < HN_G; ++w) {
            unsigned long long m = live_m[w];
            while (m && nt < HN_G) {
                const int idx = w * 64 + (int)__builtin_ctzll(m);
                m &= m - 1ull;
#pragma unroll
                for (int g = 0; g < HN_G; ++g) if (g == nt) tg[g] = idx;
                ++nt;
            }
        }
        if (nt == 0) break;                            // workgroup-uniform
        if (nt > post - kept) nt = post - kept;
        if (!dead && tid > tg[0]) {
            unsigned gm = 0u;                          // tentatives this candidate has to be clipped against
#pragma unroll
            for (int g = 0; g < HN_G; ++g) {
                const int i = tg[g];
                if (g >= nt || tid <= i) continue;
                // standup-box prefilter (iou_jit, eps = 0): overlap must be strictly positive
                const float iw = fminf(sb_s[i][2], sb_s[tid][2]) - fmaxf(sb_s[i][0], sb_s[tid][0]);
                const float ih = fminf(sb_s[i][3], sb_s[tid][3]) - fmaxf(sb_s[i][1], sb_s[tid][1]);
                if (iw > 0.f && ih > 0.f) {
                    // Bounds that cannot change the outcome but spare exact clips: the intersection is at most the smaller
                    // box, the union at least the larger, so IoU <= min / max of the two areas; the intersection is at most
                    // the overlap of the standup boxes, so IoU <= o / (A + B - o); and at most the overlap in either box's
                    // own frame.  All with a 0.1 % margin for their own rounding; NaN / inf areas compare false here exactly
                    // where the tests after the clip would.
                    const float ai = area_s[i], at = area_s[tid], o = iw * ih;
                    bool maybe = fminf(ai, at) >= lim * fmaxf(ai, at) && o >= lim * (ai + at - o);
                    if (maybe) maybe = obb_bound(cx_s[i], cy_s[i], ai, cx_s[tid], cy_s[tid], at, lim) &&
                                       obb_bound(cx_s[tid], cy_s[tid], at, cx_s[i], cy_s[i], ai, lim);
                    if (maybe) gm |= 1u << g;
                }
            }
            if (gm) work_s[atomicAdd(&s_wcnt, 1)] = (int)((unsigned)tid | (gm << 16));      // one entry per candidate: <= n <= HN_MAXK
        }
        __syncthreads();                               // the super-round's work list is complete
        const int c = s_wcnt;
        // one (entry, tentative) slot per thread, so that a candidate that meets several tentatives has its clips side by
        // side in different lanes instead of one after the other in one lane
        for (int w = tid; w < c * HN_G; w += HN_THREADS) {
            const int pr = work_s[w / HN_G], j = pr & 0xffff, g = w % HN_G;
            if (!(((unsigned)pr >> (16 + g)) & 1u)) continue;
            int i = tg[0];
#pragma unroll
            for (int h = 1; h < HN_G; ++h) if (h == g) i = tg[h];
#if defined(AL3D_NMS_STOP) && AL3D_NMS_STOP == 3
            const float inter = 0.f;
#else
            const float inter = quad_intersection_area(cx_s[i], cy_s[i], cx_s[j], cy_s[j]);
#endif
            if (inter > 0.f) {
                const float uni = area_s[i] + area_s[j] - inter;
                if (uni > 0.f && inter / uni >= p.iou_thresh) atomicOr(&supm[j / HN_GPW], (1u << g) << HN_GSH(j));
            }
        }
        __syncthreads();                               // the verdicts are visible
        unsigned standing = 1u;                        // t_0 always stands
#pragma unroll
        for (int g = 1; g < HN_G; ++g) {
            if (g >= nt) continue;
            const unsigned mg = (supm[tg[g] / HN_GPW] >> HN_GSH(tg[g])) & HN_GMASK;
            if (!(mg & standing)) standing |= 1u << g;
        }
        const unsigned mine = (supm[tid / HN_GPW] >> HN_GSH(tid)) & HN_GMASK;
        if (mine & standing) dead = true;
#pragma unroll
        for (int g = 0; g < HN_G; ++g) if (g < nt && tid == tg[g]) dead = true;      // kept or fallen: no longer a candidate
        if (tid == 0) {
            int k2 = kept;
#pragma unroll
            for (int g = 0; g < HN_G; ++g) if (g < nt && ((standing >> g) & 1u)) keep_s[k2++] = tg[g];
        }
        kept += __builtin_popcount(standing);
        __syncthreads();                               // everybody has read supm / s_wcnt
        if (tid % HN_GPW == 0) supm[tid / HN_GPW] = 0u;
        if (tid == 0) s_wcnt = 0;
    }
    if (tid == 0) s_kept = kept;
    __syncthreads();
    // ---- write survivors (kept order), applying the centre range mask; compact in order
    kept = s_kept;
    __shared__ int pass_s[128];
    if (tid < 128) pass_s[tid] = 0;
    __syncthreads();
    // each kept box is owned by the thread that decoded it (tid == candidate rank)
    bool mine = false; int kpos = -1;
    for (int q = 0; q < kept; ++q) if (keep_s[q] == tid) { mine = true; kpos = q; }
    bool ok = false;
    if (mine) {
        ok = box[0] >= p.range[0] && box[1] >= p.range[1] && box[2] >= p.range[2] &&
             box[0] <= p.range[3] && box[1] <= p.range[4] && box[2] <= p.range[5];
        pass_s[kpos] = ok ? 1 : 0;
    }
    __syncthreads();
    if (mine && ok) {
        int dst = 0;
        for (int q = 0; q < kpos; ++q) dst += pass_s[q];
        const int64_t o = ((int64_t)b * p.ntasks + t) * p.post_max + dst;
#pragma unroll
        for (int k = 0; k < 9; ++k) p.boxes[o * 9 + k] = box[k];
        p.scores[o] = my_score;
        p.labels[o] = my_label + tk.label_off;
    }
    if (tid == 0) {
        int c = 0;
        for (int q = 0; q < kept; ++q) c += pass_s[q];
        p.counts[b * p.ntasks + t] = c;
    }
}

extern "C" int64_t al3d_head_decode_nms_workspace_bytes(int B, int ntasks, const int* task_A)
{
    int64_t a = 0;
    for (int t = 0; t < ntasks; ++t) a += task_A[t];
    return (a * (B > 0 ? B : 1) + 1) * 4;
}

extern "C" int al3d_head_decode_nms(const float* hout, int B, int HW, int CH, int ntasks,
                                    const float* const* anchors, const int* task_A, const int* task_na,
                                    const int* task_nc, const int* box_off, const int* cls_off,
                                    const int* label_off, float score_thresh, float iou_thresh,
                                    int pre_max, int post_max, const float* range6, float* boxes,
                                    float* scores, int* labels, int* counts, void* workspace, void* stream)
{
    AL3D_REQUIRE(hout && anchors && task_A && task_na && task_nc && box_off && cls_off && label_off &&
                     range6 && boxes && scores && labels && counts, "al3d_head_decode_nms: null pointer");
    AL3D_REQUIRE(ntasks >= 1 && ntasks <= 8, "al3d_head_decode_nms: ntasks must be in [1,8]");
    AL3D_REQUIRE(pre_max >= 1 && pre_max <= HN_MAXK, "al3d_head_decode_nms: pre_max must be in [1,%d]", HN_MAXK);
    AL3D_REQUIRE(post_max >= 1 && post_max <= 128, "al3d_head_decode_nms: post_max must be in [1,128]");
    HeadParams p;
    p.hout = hout; p.B = B; p.HW = HW; p.CH = CH; p.ntasks = ntasks;
    p.score_thresh = score_thresh; p.iou_thresh = iou_thresh; p.pre_max = pre_max; p.post_max = post_max;
    for (int k = 0; k < 6; ++k) p.range[k] = range6[k];
    for (int t = 0; t < ntasks; ++t) {
        AL3D_REQUIRE(task_A[t] == HW * task_na[t], "al3d_head_decode_nms: task %d anchor count mismatch", t);
        p.task[t].anchors = anchors[t]; p.task[t].A = task_A[t]; p.task[t].na = task_na[t];
        p.task[t].nc = task_nc[t]; p.task[t].box_off = box_off[t]; p.task[t].cls_off = cls_off[t];
        p.task[t].label_off = label_off[t];
        AL3D_REQUIRE(box_off[t] + task_na[t] * 10 <= CH && cls_off[t] + task_na[t] * task_nc[t] <= CH,
                     "al3d_head_decode_nms: task %d channel window exceeds CH", t);
    }
    p.boxes = boxes; p.scores = scores; p.labels = labels; p.counts = counts;
    AL3D_REQUIRE(workspace, "al3d_head_decode_nms: null workspace");
    p.sbits = (unsigned*)workspace;
    int64_t soff = 0;
    int amax = 0;
    for (int t = 0; t < ntasks; ++t) { p.task_soff[t] = soff; soff += task_A[t]; if (task_A[t] > amax) amax = task_A[t]; }
    p.sample_stride = soff;
    for (int t = 0; t < 8; ++t) p.order[t] = t;
    for (int a = 1; a < ntasks; ++a)                   // insertion sort by anchor count, descending, stable
        for (int q = a; q > 0 && task_A[p.order[q]] > task_A[p.order[q - 1]]; --q) {
            const int tmp = p.order[q]; p.order[q] = p.order[q - 1]; p.order[q - 1] = tmp;
        }
    int cls_lo = CH, cls_hi = 0;
    for (int t = 0; t < ntasks; ++t) {
        if (cls_off[t] < cls_lo) cls_lo = cls_off[t];
        if (cls_off[t] + task_na[t] * task_nc[t] > cls_hi) cls_hi = cls_off[t] + task_na[t] * task_nc[t];
    }
    if (cls_hi - cls_lo <= HS_MAXW)      // all class logits of a location in one window of its record: staged pre-pass
        hipLaunchKernelGGL(head_score_rows_kernel, dim3((unsigned)al3d_cdiv(HW, HS_LOCS), (unsigned)B), dim3(256),
                           (size_t)HS_LOCS * (cls_hi - cls_lo + 1) * sizeof(float), (hipStream_t)stream, p, cls_lo, cls_hi - cls_lo);
    else
        hipLaunchKernelGGL(head_score_kernel, dim3((unsigned)al3d_cdiv(amax, 256), (unsigned)ntasks, (unsigned)B), dim3(256),
                           0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(head_nms_kernel, dim3((unsigned)(B * ntasks)), dim3(HN_THREADS), 0,
                       (hipStream_t)stream, p);
    AL3D_CHECK_LAUNCH("head_nms_kernel");
    return AL3D_OK;
}

// Stand-alone decode (GroundBox3dCoderTorch.decode_torch, det3d/core/bbox/box_coders.py:106-109).
__global__ void box_decode_kernel(const float* __restrict__ enc, const float* __restrict__ anc, int64_t n,
                                  float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[9];
    decode_box(enc + 10 * i, anc + 9 * i, o);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[9 * i + k] = o[k];
}

extern "C" int al3d_box_decode_f32(const float* enc, const float* anchors, int64_t n, float* out,
                                   void* stream)
{
    AL3D_REQUIRE(enc && anchors && out && n >= 0, "al3d_box_decode_f32: bad arguments");
    if (n == 0) return AL3D_OK;
    hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)al3d_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, enc, anchors, n, out);
    AL3D_CHECK_LAUNCH("box_decode_kernel");
    return AL3D_OK;
}
