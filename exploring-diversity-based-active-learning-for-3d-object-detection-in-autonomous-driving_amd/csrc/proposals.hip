// TransFusion query initialisation on device (bevfusion/mmdet3d/models/heads/bbox/transfusion.py:236-275): from the dense
// class heat map of a BEV grid pick the num_proposals best (class, cell) pairs among the local maxima and build the
// decoder's initial queries.
//   reference:  heatmap = sigmoid(dense_heatmap); local_max = max_pool2d(heatmap, k, stride 1) on the interior (the k//2-wide
//               frame never proposes); classes of small objects (nuScenes 8, 9; Waymo 1, 2) keep every cell;
//               heatmap *= (heatmap == local_max); top = heatmap.view(B, -1).argsort(descending)[:, :P];
//               class = top // HW, cell = top % HW; query_feat = lidar_feat[cell] + class_encoding(one_hot(class));
//               query_pos = bev_pos[cell]; query_heatmap_score = heatmap[:, :, cell].
// Here: (1) one thread per cell evaluates the sigmoid of its C logits and of the ring around it and writes the masked scores
// as order-preserving 32-bit keys [B][C * HW] (a non-negative float's bits sort like the float); (2) one workgroup per
// sample selects the P largest keys by a four-pass radix select over LDS histograms, breaks ties by the smaller flat index
// (the reference's argsort leaves ties unspecified), sorts the P winners by (score descending, index ascending) and writes
// class, cell, the C masked scores of each winning cell, the query position and the query feature row (gathered token row +
// the class-encoding column + bias).  No library kernel between the heat-map convolution and the decoder.
#include "al3d_common.h"

#define TP_THREADS 1024
#define TP_MAXP 256

__device__ __forceinline__ float tp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// logits [B][H][W][C] (channels-last) -> keys [B][C][H*W] (bits of the masked sigmoid score; 0 = not a local maximum)
__global__ __launch_bounds__(256) void tp_peak_kernel(const float* __restrict__ logits, int B, int H, int W, int C, int k,
                                                      unsigned free_mask, unsigned* __restrict__ keys)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)B * H * W) return;
    const int x = (int)(t % W), y = (int)((t / W) % H), b = (int)(t / ((int64_t)W * H));
    const int r = k / 2;
    const bool interior = y >= r && y < H - r && x >= r && x < W - r;
    const float* row = logits + ((int64_t)b * H * W) * C;
    for (int c = 0; c < C; ++c) {
        const float s = tp_sigmoid(row[((int64_t)y * W + x) * C + c]);
        bool peak = r == 0 || ((free_mask >> c) & 1u);
        if (!peak && interior) {
            peak = true;
            for (int dy = -r; dy <= r && peak; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    if (!dy && !dx) continue;
                    if (tp_sigmoid(row[((int64_t)(y + dy) * W + (x + dx)) * C + c]) > s) { peak = false; break; }
                }
        }
        keys[((int64_t)b * C + c) * H * W + (int64_t)y * W + x] = peak ? __float_as_uint(s) : 0u;
    }
}

__device__ __forceinline__ int tp_block_exclusive_scan(int v, int* s_wave, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    if (wave == 0) {
        int w = lane < TP_THREADS / 64 ? s_wave[lane] : 0;
        for (int off = 1; off < TP_THREADS / 64; off <<= 1) {
            const int y = __shfl_up(w, off);
            if (lane >= off) w += y;
        }
        if (lane < TP_THREADS / 64) s_wave[lane] = w;
    }
    __syncthreads();
    const int base = wave > 0 ? s_wave[wave - 1] : 0;
    total = s_wave[TP_THREADS / 64 - 1];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(TP_THREADS) void tp_select_kernel(const unsigned* __restrict__ keys, int HW, int C, int P,
                                                               const float* __restrict__ tokens, int hidden,
                                                               const float* __restrict__ bev_pos,
                                                               const float* __restrict__ class_cols,     // [C][hidden]
                                                               const float* __restrict__ class_bias,     // [hidden]
                                                               int64_t* __restrict__ top_class, int64_t* __restrict__ top_cell,
                                                               float* __restrict__ qscore, float* __restrict__ qfeat,
                                                               float* __restrict__ qpos)
{
    __shared__ unsigned hist[256];
    __shared__ int s_wave[TP_THREADS / 64];
    __shared__ unsigned s_prefix, s_need;
    __shared__ unsigned long long s_sel[TP_MAXP];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int64_t N = (int64_t)C * HW;
    const unsigned* kb = keys + (int64_t)b * N;
    // ---- the P-th largest key: radix select, most significant byte first
    unsigned prefix = 0u, need = (unsigned)P;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned hi_mask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
        for (int64_t i = tid; i < N; i += TP_THREADS) {
            const unsigned v = kb[i];
            if ((v & hi_mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0u;
            int d = 255;
            for (; d > 0; --d) {
                if (acc + hist[d] >= need) break;
                acc += hist[d];
            }
            s_prefix = prefix | ((unsigned)d << shift);
            s_need = need - acc;
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        __syncthreads();
    }
    const unsigned T = prefix;                       // the threshold key; `need` of the keys equal to T are taken (lowest indices)
    // ---- collect: every key > T, and the first `need` keys == T in index order.  A thread owns a contiguous index chunk,
    // so exclusive scans of the per-thread counts give positions in index order
    const int64_t chunk = (N + TP_THREADS - 1) / TP_THREADS;
    const int64_t i0 = (int64_t)tid * chunk, i1 = i0 + chunk < N ? i0 + chunk : N;
    int ngt = 0, neq = 0;
    for (int64_t i = i0; i < i1; ++i) {
        const unsigned v = kb[i];
        ngt += v > T;
        neq += v == T;
    }
    int tot_gt, tot_eq;
    int pgt = tp_block_exclusive_scan(ngt, s_wave, tot_gt);
    int peq = tp_block_exclusive_scan(neq, s_wave, tot_eq);
    for (int i = tid; i < TP_MAXP; i += TP_THREADS) s_sel[i] = 0ull;          // padding sorts last (key 0, index "infinity")
    __syncthreads();
    for (int64_t i = i0; i < i1; ++i) {
        const unsigned v = kb[i];
        int slot = -1;
        if (v > T) slot = pgt++;
        else if (v == T) { if (peq < (int)need) slot = tot_gt + peq; ++peq; }
        // composite sort word: key descending, index ascending  ->  (key << 32) | ~index, sorted descending
        if (slot >= 0 && slot < TP_MAXP) s_sel[slot] = ((unsigned long long)v << 32) | (unsigned)(~(unsigned)i);
    }
    __syncthreads();
    // ---- bitonic sort of TP_MAXP words, descending
    for (int k2 = 2; k2 <= TP_MAXP; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            if (tid < TP_MAXP) {
                const int ixj = tid ^ j;
                if (ixj > tid) {
                    const unsigned long long a = s_sel[tid], c2 = s_sel[ixj];
                    const bool up = (tid & k2) == 0;                        // descending blocks where (tid & k2) == 0
                    if (up ? a < c2 : a > c2) { s_sel[tid] = c2; s_sel[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // ---- outputs
    for (int p = tid; p < P; p += TP_THREADS) {
        const unsigned idx = ~(unsigned)(s_sel[p] & 0xffffffffull);
        top_class[(int64_t)b * P + p] = idx / HW;
        top_cell[(int64_t)b * P + p] = idx % HW;
    }
    for (int e = tid; e < P * C; e += TP_THREADS) {
        const int c = e / P, p = e % P;
        const unsigned idx = ~(unsigned)(s_sel[p] & 0xffffffffull);
        qscore[((int64_t)b * C + c) * P + p] = __uint_as_float(kb[(int64_t)c * HW + idx % HW]);
    }
    for (int e = tid; e < P * 2; e += TP_THREADS) {
        const int p = e >> 1;
        const unsigned idx = ~(unsigned)(s_sel[p] & 0xffffffffull);
        qpos[((int64_t)b * P + p) * 2 + (e & 1)] = bev_pos[(int64_t)(idx % HW) * 2 + (e & 1)];
    }
    for (int e = tid; e < P * hidden; e += TP_THREADS) {
        const int p = e / hidden, h = e % hidden;
        const unsigned idx = ~(unsigned)(s_sel[p] & 0xffffffffull);
        const int cls = idx / HW, cell = idx % HW;
        // key_rows[cell] + class_encoding.weight[:, cls] + bias, in that order (the reference adds the Conv1d output)
        qfeat[((int64_t)b * P + p) * hidden + h] =
            tokens[((int64_t)b * HW + cell) * hidden + h] + (class_cols[(int64_t)cls * hidden + h] + class_bias[h]);
    }
}

extern "C" int64_t al3d_tf_proposals_workspace_bytes(int B, int H, int W, int C)
{
    return al3d_align((int64_t)(B > 0 ? B : 1) * H * W * C * 4, 256);
}

extern "C" int al3d_tf_proposals_f32(const float* heat_logits, int B, int H, int W, int C, int nms_kernel, unsigned free_class_mask,
                                     int P, const float* tokens, int hidden, const float* bev_pos, const float* class_cols,
                                     const float* class_bias, void* workspace, int64_t* top_class, int64_t* top_cell,
                                     float* query_heatmap_score, float* query_feat, float* query_pos, void* stream)
{
    AL3D_REQUIRE(B >= 0 && H > 0 && W > 0 && C >= 1 && C <= 32 && P >= 1 && P <= TP_MAXP && hidden >= 1,
                 "al3d_tf_proposals_f32: bad sizes (1 <= classes <= 32, 1 <= proposals <= 256)");
    AL3D_REQUIRE(nms_kernel >= 1 && (nms_kernel & 1) && nms_kernel / 2 < H && nms_kernel / 2 < W, "al3d_tf_proposals_f32: odd nms kernel");
    AL3D_REQUIRE((int64_t)H * W * C < (1ll << 31) && (int64_t)H * W * C >= P, "al3d_tf_proposals_f32: grid too large / fewer cells than proposals");
    if (B == 0) return AL3D_OK;
    AL3D_REQUIRE(heat_logits && tokens && bev_pos && class_cols && class_bias && workspace && top_class && top_cell &&
                 query_heatmap_score && query_feat && query_pos, "al3d_tf_proposals_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned* keys = (unsigned*)workspace;
    hipLaunchKernelGGL(tp_peak_kernel, dim3((unsigned)al3d_cdiv((int64_t)B * H * W, 256)), dim3(256), 0, s, heat_logits, B, H, W, C,
                       nms_kernel, free_class_mask, keys);
    hipLaunchKernelGGL(tp_select_kernel, dim3((unsigned)B), dim3(TP_THREADS), 0, s, keys, H * W, C, P, tokens, hidden, bev_pos,
                       class_cols, class_bias, top_class, top_cell, query_heatmap_score, query_feat, query_pos);
    AL3D_CHECK_LAUNCH("al3d_tf_proposals_f32");
    return AL3D_OK;
}
